// Context shared by the C-ABI entry points: one HIP stream, growable device buffers, last-error text.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>
#include "../../include/nrs.h"

namespace nrs {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct Engine;       // nrs_engine.hip
struct KltState;     // nrs_klt.hip
struct ShiState;     // nrs_shi.hip

// Exchange steps of a sharded solve (nrs_comm.hip): one process per GPU, every call is ordered on the
// context's stream.  Two primitives are all the engine needs:
//   allreduce : element-wise sum of n doubles over the ranks, the same bits on every rank
//   exchange  : boundary rows of a replicated-layout row vector with rank-1 / rank+1 (offsets in doubles)
struct HaloPlan { size_t lo_send = 0, lo_send_n = 0, hi_send = 0, hi_send_n = 0, lo_recv = 0, lo_recv_n = 0, hi_recv = 0, hi_recv_n = 0; };
struct Comm {
    int rank = 0, world = 1;
    virtual ~Comm() {}
    virtual int allreduce(nrs_ctx* c, const double* send, double* recv, size_t n) = 0;
    virtual int exchange(nrs_ctx* c, double* vec, const HaloPlan& h, hipStream_t stream) = 0;   // ordered on `stream`
};
struct Arena { char* base = nullptr; size_t cap = 0, off = 0; };

// Debug / A-B switches (include/nrs.h "Debug switches").  A context takes the NRS_* variables of the environment ONCE, when it is created
// (plus NRS_DEBUG="NAME=VALUE,NAME2=VALUE2", names without the prefix), and is changed afterwards through nrs_debug_set only: the library
// never looks at the environment again -- a host application may call setenv at any time, and no launch path pays for a lookup (the list
// is empty in normal use).  Entry points without a context (the host edge builder) read a process-wide snapshot taken on first use.
struct DebugOpts {
    std::vector<std::pair<std::string, std::string>> kv;
    const char* get(const char* name) const {
        for (const auto& p : kv) if (p.first == name) return p.second.c_str();
        return nullptr;
    }
    void set(const char* name, const char* value) {
        for (size_t i = 0; i < kv.size(); ++i)
            if (kv[i].first == name) { if (value) kv[i].second = value; else kv.erase(kv.begin() + i); return; }
        if (value) kv.emplace_back(name, value);
    }
    void load_environment();         // nrs_pose_only.hip: the one place that reads the environment
    static const DebugOpts& process();
};

}  // namespace nrs

struct nrs_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipDeviceProp_t prop;
    nrs_options opt;
    nrs::DebugOpts dbg;              // debug / A-B switches: read once at nrs_create, changed by nrs_debug_set only
    const char* env(const char* name) const { return dbg.get(name); }
    char err[512];
    nrs_profile prof;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // scratch for a1
    nrs::DevBuf po_uv, po_X, po_err, po_level, po_out, po_trace, po_multi;   // (po_multi: state + partial slots of the multi-workgroup form)
    // resident BA problem (a3) and per-call tracking problems (a2): one reusable arena each
    nrs::Engine* dba = nullptr;
    nrs::Arena arena_dba, arena_trk;
    nrs::KltState* klt = nullptr;
    nrs::ShiState* shi = nullptr;
    nrs::Comm* comm = nullptr;       // set by nrs_comm_init_*: BA problems uploaded afterwards are sharded over its ranks
    hipStream_t comm_stream = nullptr;   // boundary-row exchanges run here, next to the interior tiles on `stream`
    hipEvent_t ev_vec = nullptr, ev_halo = nullptr;
    nrs::DevBuf tap;                 // residual taps (edge lists by vertex + outputs) of the engine `tap_serial`: built on first use
    unsigned long long tap_serial = 0, engine_serial = 0;
    nrs::DevBuf pack_ws, pack_ws2, pack_ws3, pack_ws4;   // device-side problem construction (nrs_engine_devpack.hpp): raw inputs + intermediates
    nrs::DevBuf dba_skin;            // ... and of the resident BA window (N2b)
    nrs::DevBuf dba_kft;             // embedded BA window: the keyframe-block factorisation (nrs_engine_kft.hpp)
    nrs::DevBuf nd_skin;             // embedded mode (nrs_engine_skin.hpp): the skinned observations of the tracking engine
    void* plan_worker = nullptr;     // nrs_engine_nd.hpp PlanWorker: the helper thread of the direct solver's symbolic phase (one for the context's lifetime)
    void* nd_cache = nullptr;        // direct solver of the tracking engines (nrs_engine_nd.hpp NdCache): the last few plans with their device arrays
    nrs::DevBuf comm_flag;           // one double: status word the ranks agree on after a sharded upload
    nrs::DevBuf gather_ws;           // sharded download / taps: two full-length row vectors for the gather (a rank holds its own rows only); released after use
    bool err_local = true;           // last set-up failure may be specific to this rank (allocation, HIP, rank-dependent checks)
    double* pin_scal = nullptr;      // pinned host mirrors of the engine's scalars / flags
    int* pin_flags = nullptr;
    int seq = 0;                     // sequence number of the last publication the host waited for (pin_flags[7])
    // speculative LM trials of the single-frame engines (nrs_engine_types.hpp SpecSet): mirrors, streams and events of the shadow sets
    double* pin_spec_scal = nullptr; int* pin_spec_flags = nullptr;
    int spec_run = 4;                // rejections of the last completed run of an LM iteration (sizes the next batch)
    hipStream_t spec_stream[3] = {nullptr, nullptr, nullptr};
    hipEvent_t spec_fork = nullptr, spec_join[3] = {nullptr, nullptr, nullptr};
    hipEvent_t spec_back[4] = {nullptr, nullptr, nullptr, nullptr};   // behind the back pass of the batch's k-th trial (they run one after the other: nd_solve_enqueue)

    int fail(int code, const char* fmt, ...) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(err, sizeof(err), fmt, ap);
        va_end(ap);
        return code;
    }
    int ensure(nrs::DevBuf& b, size_t bytes) {
        if (bytes <= b.cap && b.p) return NRS_OK;
        if (b.p) (void)hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&b.p, want);
        if (e != hipSuccess) return fail(NRS_ERR_ALLOC, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        b.cap = want;
        if (env("NRS_POISON")) { (void)hipMemset(b.p, 0xFF, want); (void)hipDeviceSynchronize(); }     // (debug: a read of memory nobody wrote shows up as NaN)
        return NRS_OK;
    }
    void release(nrs::DevBuf& b) {
        if (b.p) (void)hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    }
};

#define NRS_HIP(ctx, call)                                                                     \
    do {                                                                                       \
        hipError_t e__ = (call);                                                               \
        if (e__ != hipSuccess)                                                                 \
            return (ctx)->fail(NRS_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                               __FILE__, __LINE__);                                            \
    } while (0)

#define NRS_TRY(expr)                \
    do {                             \
        int rc__ = (expr);           \
        if (rc__ != NRS_OK) return rc__; \
    } while (0)

namespace nrs {
void dba_free(nrs_ctx* ctx);
void nd_cache_free(nrs_ctx* ctx);
void comm_free(nrs_ctx* ctx);
int comm_agree(nrs_ctx* ctx, int rc);    // collective: 0 if every rank passed 0, else an error on every rank
void klt_free(nrs_ctx* ctx);
void shi_free(nrs_ctx* ctx);
}
