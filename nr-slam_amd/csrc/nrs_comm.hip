// Exchange steps of a sharded bundle-adjustment solve: one process per GPU, the ranks
// own contiguous ranges of keyframes, and per linearisation / PCG iteration they exchange
//   * an all-reduce (sum) of the pose-side normal-equation blocks and the scalar sums, and
//   * the rows of the two boundary keyframes with the neighbouring ranks (dampers reach one keyframe
//     to either side: reference g2o_optimization.cc:1073-1132 connects consecutive keyframes only).
// Two back ends behind nrs::Comm:
//   RCCL   ncclAllReduce / grouped ncclSend+ncclRecv on the context's stream (xGMI between the GPUs of a
//          node).  librccl is bound at run time with dlopen/dlsym -- the copy already loaded in the
//          process (PyTorch's) if there is one -- so single-GPU users of libnrs_hip.so do not need it.
//   local  ranks are threads of ONE process driving contexts on the same GPU: a test harness that runs
//          the sharded arithmetic on a 1-GPU box (tests/test_gpu_sharded.py); rendezvous by a host barrier.
#include <condition_variable>
#include <dlfcn.h>
#include <mutex>
#include <new>
#include <rccl/rccl.h>
#include "nrs_ctx.hpp"

namespace nrs {

// ------------------------------------------------------------------------------------- RCCL
struct RcclApi {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static const RcclApi* rccl_api(char* err, size_t errlen) {
    static RcclApi api;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (api.h) return &api;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;      // the copy the process already uses
    if (!h) for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) { snprintf(err, errlen, "librccl not found: %s", dlerror()); return nullptr; }
    RcclApi a;
    a.h = h;
#define NRS_SYM(field, name)                                                          \
    *(void**)(&a.field) = dlsym(h, name);                                             \
    if (!a.field) { snprintf(err, errlen, "librccl: missing symbol %s", name); return nullptr; }
    NRS_SYM(GetUniqueId, "ncclGetUniqueId")
    NRS_SYM(CommInitRank, "ncclCommInitRank")
    NRS_SYM(CommDestroy, "ncclCommDestroy")
    NRS_SYM(AllReduce, "ncclAllReduce")
    NRS_SYM(Send, "ncclSend")
    NRS_SYM(Recv, "ncclRecv")
    NRS_SYM(GroupStart, "ncclGroupStart")
    NRS_SYM(GroupEnd, "ncclGroupEnd")
    NRS_SYM(GetErrorString, "ncclGetErrorString")
#undef NRS_SYM
    api = a;
    return &api;
}

#define NRS_NCCL(ctx, call)                                                                         \
    do {                                                                                            \
        ncclResult_t r__ = (call);                                                                  \
        if (r__ != ncclSuccess)                                                                     \
            return (ctx)->fail(NRS_ERR_COMM, "%s failed: %s", #call, api->GetErrorString(r__));     \
    } while (0)

struct RcclComm : Comm {
    const RcclApi* api = nullptr;
    ncclComm_t comm = nullptr;
    ~RcclComm() override { if (comm) (void)api->CommDestroy(comm); }
    int allreduce(nrs_ctx* c, const double* send, double* recv, size_t n) override {
        NRS_NCCL(c, api->AllReduce(send, recv, n, ncclDouble, ncclSum, comm, c->stream));
        return NRS_OK;
    }
    int exchange(nrs_ctx* c, double* v, const HaloPlan& h, hipStream_t st) override {
        if (world == 1) return NRS_OK;
        NRS_NCCL(c, api->GroupStart());
        if (rank > 0) {
            NRS_NCCL(c, api->Send(v + h.lo_send, h.lo_send_n, ncclDouble, rank - 1, comm, st));
            NRS_NCCL(c, api->Recv(v + h.lo_recv, h.lo_recv_n, ncclDouble, rank - 1, comm, st));
        }
        if (rank < world - 1) {
            NRS_NCCL(c, api->Send(v + h.hi_send, h.hi_send_n, ncclDouble, rank + 1, comm, st));
            NRS_NCCL(c, api->Recv(v + h.hi_recv, h.hi_recv_n, ncclDouble, rank + 1, comm, st));
        }
        NRS_NCCL(c, api->GroupEnd());
        return NRS_OK;
    }
};

// ------------------------------------------------------------------------------------- local (threads, one GPU)
constexpr int LOCAL_MAX = 8;
struct LocalGroup {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int waiting = 0;
    uint64_t generation = 0;
    const double* slot[LOCAL_MAX] = {};
    // A rank that fails between two barriers poisons the group: every waiter is released with an error and every later barrier
    // fails at once.  The poison is STICKY: it is lifted only when all `world` ranks have re-joined (nrs_comm_init_local) after
    // it -- a rank that re-joins early and starts a collective while a peer is still inside the failed window gets NRS_ERR_COMM
    // instead of pairing its barriers with that peer's (different buffer lengths: k_local_sum would read past the shorter one).
    bool aborted = false;
    bool rejoined[LOCAL_MAX] = {};   // ranks that have re-joined since the abort
    bool barrier() {
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) return false;
        const uint64_t g = generation;
        if (++waiting == world) { waiting = 0; ++generation; cv.notify_all(); return true; }
        cv.wait(lk, [&] { return generation != g || aborted; });
        if (generation == g) { --waiting; return false; }          // released by an abort: this waiter leaves the barrier it never completed
        return true;
    }
    void abort() {
        std::lock_guard<std::mutex> lk(mu);
        if (!aborted) for (bool& r : rejoined) r = false;
        aborted = true;
        cv.notify_all();
    }
    void join(int rank) {            // nrs_comm_init_local: a rank (re)joins
        std::lock_guard<std::mutex> lk(mu);
        if (!aborted) return;
        rejoined[rank] = true;
        bool all = waiting == 0;
        for (int r = 0; r < world; ++r) all = all && rejoined[r];
        if (all) { aborted = false; ++generation; }
    }
};

// a failing step of a thread rank must not leave its peers in a barrier
#define NRS_LOCAL(ctx, grp, expr)                                                      \
    do {                                                                               \
        int rc__ = (expr);                                                             \
        if (rc__ != NRS_OK) { (grp)->abort(); return rc__; }                           \
    } while (0)
#define NRS_LOCAL_BARRIER(ctx, grp)                                                    \
    do {                                                                               \
        if (!(grp)->barrier()) return (ctx)->fail(NRS_ERR_COMM, "a peer rank of the local group failed"); \
    } while (0)
static inline int hip_rc(nrs_ctx* c, hipError_t e, const char* what) {
    return e == hipSuccess ? NRS_OK : c->fail(NRS_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
}

struct PtrPack { const double* p[LOCAL_MAX]; };

__global__ void k_local_sum(PtrPack in, int world, double* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0;
    for (int r = 0; r < world; ++r) s += in.p[r][i];               // rank order: the same bits on every rank
    out[i] = s;
}

struct LocalComm : Comm {
    LocalGroup* g = nullptr;
    DevBuf tmp;
    ~LocalComm() override { if (tmp.p) (void)hipFree(tmp.p); }
    int allreduce(nrs_ctx* c, const double* send, double* recv, size_t n) override {
        g->slot[rank] = send;
        NRS_LOCAL(c, g, hip_rc(c, hipStreamSynchronize(c->stream), "hipStreamSynchronize"));
        NRS_LOCAL_BARRIER(c, g);
        PtrPack pp;
        for (int r = 0; r < LOCAL_MAX; ++r) pp.p[r] = r < world ? g->slot[r] : nullptr;
        NRS_LOCAL(c, g, c->ensure(tmp, sizeof(double) * n));       // send == recv is allowed: sum into a staging buffer
        hipLaunchKernelGGL(k_local_sum, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, pp, world, tmp.as<double>(), n);
        NRS_LOCAL(c, g, hip_rc(c, hipStreamSynchronize(c->stream), "k_local_sum"));
        NRS_LOCAL_BARRIER(c, g);                                   // everybody has read every send buffer
        NRS_LOCAL(c, g, hip_rc(c, hipMemcpyAsync(recv, tmp.p, sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream), "hipMemcpyAsync"));
        return NRS_OK;
    }
    int exchange(nrs_ctx* c, double* v, const HaloPlan& h, hipStream_t st) override {
        if (world == 1) return NRS_OK;
        g->slot[rank] = v;
        NRS_LOCAL(c, g, hip_rc(c, hipStreamSynchronize(st), "hipStreamSynchronize"));
        NRS_LOCAL_BARRIER(c, g);
        // the layout is the same on every rank: a neighbour's send range is this rank's receive range
        if (rank > 0) NRS_LOCAL(c, g, hip_rc(c, hipMemcpyAsync(v + h.lo_recv, g->slot[rank - 1] + h.lo_recv, sizeof(double) * h.lo_recv_n, hipMemcpyDeviceToDevice, st), "halo copy"));
        if (rank < world - 1) NRS_LOCAL(c, g, hip_rc(c, hipMemcpyAsync(v + h.hi_recv, g->slot[rank + 1] + h.hi_recv, sizeof(double) * h.hi_recv_n, hipMemcpyDeviceToDevice, st), "halo copy"));
        NRS_LOCAL(c, g, hip_rc(c, hipStreamSynchronize(st), "hipStreamSynchronize"));
        NRS_LOCAL_BARRIER(c, g);
        return NRS_OK;
    }
};

void comm_free(nrs_ctx* c) {
    // exchanges may still be in flight on either stream: drain both before the communicator goes away
    if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    delete c->comm;
    c->comm = nullptr;
    if (c->comm_stream) { (void)hipStreamDestroy(c->comm_stream); c->comm_stream = nullptr; }
    if (c->ev_vec) { (void)hipEventDestroy(c->ev_vec); c->ev_vec = nullptr; }
    if (c->ev_halo) { (void)hipEventDestroy(c->ev_halo); c->ev_halo = nullptr; }
}

// Every rank passes the status of its rank-local set-up; all of them return NRS_OK, or none does.  Without
// this a rank whose set-up failed returns to its caller while its peers walk into the first collective.
int comm_agree(nrs_ctx* c, int rc) {
    if (!c->comm || c->comm->world == 1) return rc;
    double flag = rc == NRS_OK ? 0.0 : 1.0, sum = 0.0;
    if (c->ensure(c->comm_flag, 2 * sizeof(double)) != NRS_OK) return rc != NRS_OK ? rc : NRS_ERR_ALLOC;
    double* d = c->comm_flag.as<double>();
    char keep[sizeof(c->err)];
    memcpy(keep, c->err, sizeof(keep));                            // the first failure's text is the useful one
    int st = NRS_OK;
    if (hipMemcpyAsync(d, &flag, sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess) st = NRS_ERR_HIP;
    if (st == NRS_OK) st = c->comm->allreduce(c, d, d + 1, 1);
    if (st == NRS_OK && hipMemcpyAsync(&sum, d + 1, sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess) st = NRS_ERR_HIP;
    if (st == NRS_OK && hipStreamSynchronize(c->stream) != hipSuccess) st = NRS_ERR_HIP;
    if (rc != NRS_OK) { memcpy(c->err, keep, sizeof(keep)); return rc; }
    if (st != NRS_OK) return st;
    if (sum != 0.0) return c->fail(NRS_ERR_COMM, "%d peer rank(s) failed to set up the sharded window", (int)sum);
    return NRS_OK;
}

// second stream + the two events that order it against the context's stream (hand-offs are events only)
static int comm_streams(nrs_ctx* c) {
    NRS_HIP(c, hipSetDevice(c->device));
    if (!c->comm_stream) NRS_HIP(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    if (!c->ev_vec) NRS_HIP(c, hipEventCreateWithFlags(&c->ev_vec, hipEventDisableTiming));
    if (!c->ev_halo) NRS_HIP(c, hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming));
    return NRS_OK;
}

}  // namespace nrs

using namespace nrs;

extern "C" int nrs_comm_unique_id(uint8_t* id, int32_t capacity) {
    if (!id || capacity < (int32_t)sizeof(ncclUniqueId)) return NRS_ERR_INVALID;
    char err[256];
    const RcclApi* api = rccl_api(err, sizeof(err));
    if (!api) return NRS_ERR_COMM;
    ncclUniqueId u;
    if (api->GetUniqueId(&u) != ncclSuccess) return NRS_ERR_COMM;
    memset(id, 0, (size_t)capacity);
    memcpy(id, &u, sizeof(u));
    return NRS_OK;
}

extern "C" int nrs_comm_init_rccl(nrs_ctx* c, int32_t world, int32_t rank, const uint8_t* id, int32_t id_bytes) {
    if (!c) return NRS_ERR_INVALID;
    if (world < 1 || rank < 0 || rank >= world || !id || id_bytes < (int32_t)sizeof(ncclUniqueId))
        return c->fail(NRS_ERR_INVALID, "nrs_comm_init_rccl: bad argument");
    if (c->dba) return c->fail(NRS_ERR_STATE, "set the communicator before uploading a problem");
    comm_free(c);
    const RcclApi* api = rccl_api(c->err, sizeof(c->err));
    if (!api) return NRS_ERR_COMM;
    NRS_HIP(c, hipSetDevice(c->device));
    RcclComm* rc = new (std::nothrow) RcclComm();
    if (!rc) return c->fail(NRS_ERR_ALLOC, "out of host memory");
    rc->api = api; rc->rank = rank; rc->world = world;
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclResult_t r = api->CommInitRank(&rc->comm, world, u, rank);
    if (r != ncclSuccess) { rc->comm = nullptr; delete rc; return c->fail(NRS_ERR_COMM, "ncclCommInitRank failed: %s", api->GetErrorString(r)); }
    c->comm = rc;
    return comm_streams(c);
}

extern "C" int nrs_local_group_create(int32_t world, void** group) {
    if (!group || world < 1 || world > LOCAL_MAX) return NRS_ERR_INVALID;
    LocalGroup* g = new (std::nothrow) LocalGroup();
    if (!g) return NRS_ERR_ALLOC;
    g->world = world;
    *group = g;
    return NRS_OK;
}

extern "C" void nrs_local_group_destroy(void* group) { delete static_cast<LocalGroup*>(group); }

extern "C" int nrs_comm_init_local(nrs_ctx* c, void* group, int32_t rank) {
    if (!c) return NRS_ERR_INVALID;
    LocalGroup* g = static_cast<LocalGroup*>(group);
    if (!g || rank < 0 || rank >= g->world) return c->fail(NRS_ERR_INVALID, "nrs_comm_init_local: bad argument");
    if (c->dba) return c->fail(NRS_ERR_STATE, "set the communicator before uploading a problem");
    comm_free(c);
    LocalComm* lc = new (std::nothrow) LocalComm();
    if (!lc) return c->fail(NRS_ERR_ALLOC, "out of host memory");
    lc->g = g; lc->rank = rank; lc->world = g->world;
    g->join(rank);
    c->comm = lc;
    return comm_streams(c);
}

extern "C" int nrs_comm_rank(const nrs_ctx* c, int32_t* rank, int32_t* world) {
    if (!c) return NRS_ERR_INVALID;
    if (rank) *rank = c->comm ? c->comm->rank : 0;
    if (world) *world = c->comm ? c->comm->world : 1;
    return NRS_OK;
}
