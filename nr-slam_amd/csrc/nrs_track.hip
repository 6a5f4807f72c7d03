// a2: CameraPoseAndDeformationOptimization (reference modules/optimization/g2o_optimization.cc:148-557)
// and the RegularizationGraph operations it embeds:
//   a19 GetEdges      (reference modules/map/regularization_graph.cc:61-87)
//   a20 UpdateVertex  (reference modules/map/regularization_graph.cc:89-146)
//
// Division of labour: every floating-point pass over points or edges (residuals, Jacobians, the
// normal equations, the PCG solves, edge re-weighting, neighbour ranking) is a HIP kernel; the host
// runs the reference's bookkeeping between them (which edges exist, levels between the two inlier
// rounds, the IQR test, statuses), because those are data-dependent container walks whose order is
// part of the reference's semantics (SURVEY.md 8a "container-order dependencies").
#include <algorithm>
#include <cmath>
#include <chrono>
#include "nrs_engine.hpp"

namespace nrs {

// ---------------------------------------------------------------------------------------------
// InterpolationWeight (utilities/geometry_toolbox.cc:26-28): float argument, exp evaluated in
// double and rounded to float -- the value a correctly rounded expf returns (include/nrs.h).
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline float interpolation_weight(float d, float sigma) {
#pragma clang fp contract(off)
    const float arg = -(d * d) / (2.0f * sigma * sigma);
    return (float)exp((double)arg);
}

// a19, step 1: rank of every directed entry inside its row under (status asc, weight desc, index asc).  One wave per row: the row's
// (status, weight) pairs are read once, 64 at a time, and compared lane against lane (a thread per row re-read them from memory
// deg^2 times: 140 us for 1k rows of ~30 entries)
__global__ __launch_bounds__(256) void k_graph_rank(int n, const int* __restrict__ rowptr, const int* __restrict__ eid,
                                                    const float* __restrict__ e_w, const int* __restrict__ e_status, int* rank) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= n) return;
    const int lo = rowptr[p], hi = rowptr[p + 1];
    for (int a0 = lo; a0 < hi; a0 += 64) {
        const int a = a0 + lane;
        int sa = 0;
        float wa = 0.f;
        if (a < hi) { const int e = eid[a]; sa = e_status[e]; wa = e_w[e]; }
        int r = 0;
        for (int b0 = lo; b0 < hi; b0 += 64) {
            const int b = b0 + lane;
            int sb = 0;
            float wb = 0.f;
            if (b < hi) { const int e = eid[b]; sb = e_status[e]; wb = e_w[e]; }
            const int nb = min(64, hi - b0);
            for (int j = 0; j < nb; ++j) {
                const int sj = __shfl(sb, j, 64);
                const float wj = __shfl(wb, j, 64);
                const bool before = (sj != sa) ? (sj < sa) : ((wj != wa) ? (wj > wa) : (b0 + j < a));
                r += before ? 1 : 0;
            }
        }
        if (a < hi) rank[a] = r;
    }
}

// a19, step 2: cut position = rank of the first entry (in sorted order) whose weight < min_weight
__global__ void k_graph_cut(int n, const int* __restrict__ rowptr, const int* __restrict__ eid,
                            const float* __restrict__ e_w, const int* __restrict__ rank, float min_w, int* count) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int lo = rowptr[p], hi = rowptr[p + 1];
    int cut = hi - lo;
    for (int a = lo; a < hi; ++a)
        if (e_w[eid[a]] < min_w) cut = min(cut, rank[a]);
    count[p] = cut;
}

// a19, step 3: scatter the kept entries to their sorted position
__global__ void k_graph_scatter(int n, const int* __restrict__ rowptr, const int* __restrict__ col,
                                const int* __restrict__ eid, const int* __restrict__ rank,
                                const int* __restrict__ o_rowptr, int* o_col, int* o_eid) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int lo = rowptr[p], hi = rowptr[p + 1];
    const int base = o_rowptr[p], cnt = o_rowptr[p + 1] - base;
    for (int a = lo; a < hi; ++a)
        if (rank[a] < cnt) { o_col[base + rank[a]] = col[a]; o_eid[base + rank[a]] = eid[a]; }
}

// a20: UpdateVertex for a list of points.  Two points sharing an edge compute the same values
// (the positions are final), so concurrent updates of one edge are idempotent.
__global__ void k_graph_update(int n_ids, const int* __restrict__ ids, const int* __restrict__ rowptr,
                               const int* __restrict__ col, const int* __restrict__ eid, const float* __restrict__ pos,
                               float* e_w, float* e_max, float* e_min, int* e_status, float sigma, float stretch_th,
                               int* good) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_ids) return;
    const int p = ids[i];
    const float px = pos[3 * p], py = pos[3 * p + 1], pz = pos[3 * p + 2];
    int n_good = 0;
    for (int a = rowptr[p]; a < rowptr[p + 1]; ++a) {
        const int o = col[a], e = eid[a];
        const float dx = px - pos[3 * o], dy = py - pos[3 * o + 1], dz = pz - pos[3 * o + 2];
        const float d = sqrtf(dx * dx + dy * dy + dz * dz);
        float mx = e_max[e], mn = e_min[e];
        if (d > mx) mx = d;
        if (d < mn) mn = d;
        e_max[e] = mx;
        e_min[e] = mn;
        e_w[e] = interpolation_weight(mx, sigma);
        if (fabsf((mx - mn) / mn) > stretch_th) e_status[e] = NRS_GRAPH_BAD;
        else ++n_good;
    }
    good[i] = n_good;
}

struct GraphDevice {            // device mirror of an nrs_graph for the duration of one call
    nrs_ctx* c;
    std::vector<void*> allocs;
    int *rowptr = nullptr, *col = nullptr, *eid = nullptr, *status = nullptr, *rank = nullptr, *count = nullptr;
    int *o_rowptr = nullptr, *o_col = nullptr, *o_eid = nullptr, *ids = nullptr, *good = nullptr;
    float *w = nullptr, *mx = nullptr, *mn = nullptr, *pos = nullptr;
    int n = 0, nnz = 0, ne = 0;
    ~GraphDevice() { for (void* p : allocs) (void)hipFree(p); }
    template <class Tp> int alloc(Tp** p, size_t n_) {
        hipError_t e = hipMalloc((void**)p, std::max<size_t>(16, n_ * sizeof(Tp)));
        if (e != hipSuccess) return c->fail(NRS_ERR_ALLOC, "hipMalloc failed: %s", hipGetErrorString(e));
        allocs.push_back(*p);
        return NRS_OK;
    }
};

static int graph_validate(nrs_ctx* c, const nrs_graph* g) {
    if (!g || g->n_points < 0 || g->n_edges < 0 || !g->rowptr) return c->fail(NRS_ERR_INVALID, "graph: null/negative");
    const int nnz = g->rowptr[g->n_points];
    if (nnz > 0 && (!g->col || !g->eid || !g->e_w || !g->e_d0 || !g->e_max || !g->e_min || !g->e_status))
        return c->fail(NRS_ERR_INVALID, "graph: null arrays");
    if (!(g->sigma > 0)) return c->fail(NRS_ERR_INVALID, "graph: sigma must be positive");
    for (int p = 0; p < g->n_points; ++p) {
        if (g->rowptr[p + 1] < g->rowptr[p]) return c->fail(NRS_ERR_INVALID, "graph: rowptr not monotone");
        for (int a = g->rowptr[p]; a < g->rowptr[p + 1]; ++a) {
            if (g->col[a] < 0 || g->col[a] >= g->n_points || g->eid[a] < 0 || g->eid[a] >= g->n_edges)
                return c->fail(NRS_ERR_INVALID, "graph: index out of range");
            if (a > g->rowptr[p] && g->col[a] <= g->col[a - 1]) return c->fail(NRS_ERR_INVALID, "graph: row not in ascending index order");
        }
    }
    return NRS_OK;
}

static int graph_upload(nrs_ctx* c, GraphDevice& G, const nrs_graph* g) {
    G.c = c;
    G.n = g->n_points;
    G.nnz = g->rowptr[g->n_points];
    G.ne = g->n_edges;
    NRS_TRY(G.alloc(&G.rowptr, G.n + 1)); NRS_TRY(G.alloc(&G.col, G.nnz)); NRS_TRY(G.alloc(&G.eid, G.nnz));
    NRS_TRY(G.alloc(&G.status, G.ne)); NRS_TRY(G.alloc(&G.w, G.ne)); NRS_TRY(G.alloc(&G.mx, G.ne)); NRS_TRY(G.alloc(&G.mn, G.ne));
    NRS_TRY(G.alloc(&G.rank, G.nnz)); NRS_TRY(G.alloc(&G.count, G.n + 1));
    NRS_TRY(G.alloc(&G.o_rowptr, G.n + 1)); NRS_TRY(G.alloc(&G.o_col, G.nnz)); NRS_TRY(G.alloc(&G.o_eid, G.nnz));
    NRS_TRY(G.alloc(&G.ids, G.n)); NRS_TRY(G.alloc(&G.good, G.n)); NRS_TRY(G.alloc(&G.pos, 3 * (size_t)G.n));
    NRS_HIP(c, hipMemcpyAsync(G.rowptr, g->rowptr, sizeof(int) * (G.n + 1), hipMemcpyHostToDevice, c->stream));
    if (G.nnz) {
        NRS_HIP(c, hipMemcpyAsync(G.col, g->col, sizeof(int) * G.nnz, hipMemcpyHostToDevice, c->stream));
        NRS_HIP(c, hipMemcpyAsync(G.eid, g->eid, sizeof(int) * G.nnz, hipMemcpyHostToDevice, c->stream));
    }
    if (G.ne) {
        NRS_HIP(c, hipMemcpyAsync(G.status, g->e_status, sizeof(int) * G.ne, hipMemcpyHostToDevice, c->stream));
        NRS_HIP(c, hipMemcpyAsync(G.w, g->e_w, sizeof(float) * G.ne, hipMemcpyHostToDevice, c->stream));
        NRS_HIP(c, hipMemcpyAsync(G.mx, g->e_max, sizeof(float) * G.ne, hipMemcpyHostToDevice, c->stream));
        NRS_HIP(c, hipMemcpyAsync(G.mn, g->e_min, sizeof(float) * G.ne, hipMemcpyHostToDevice, c->stream));
    }
    return NRS_OK;
}

// GetEdges for all points on the device copy; result on the host
static int graph_select(nrs_ctx* c, GraphDevice& G, float sigma, std::vector<int>& o_rowptr, std::vector<int>& o_col,
                        std::vector<int>& o_eid) {
    const float min_w = interpolation_weight((float)((double)sigma * 1.5), sigma);   // regularization_graph.cc:30
    const dim3 b(256), g((G.n + 255) / 256);
    o_rowptr.assign(G.n + 1, 0);
    if (G.n == 0) return NRS_OK;
    hipLaunchKernelGGL(k_graph_rank, dim3((G.n + 3) / 4), b, 0, c->stream, G.n, G.rowptr, G.eid, G.w, G.status, G.rank);
    hipLaunchKernelGGL(k_graph_cut, g, b, 0, c->stream, G.n, G.rowptr, G.eid, G.w, G.rank, min_w, G.count);
    NRS_HIP(c, hipGetLastError());
    std::vector<int> cnt(G.n);
    NRS_HIP(c, hipMemcpyAsync(cnt.data(), G.count, sizeof(int) * G.n, hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    for (int p = 0; p < G.n; ++p) o_rowptr[p + 1] = o_rowptr[p] + cnt[p];
    NRS_HIP(c, hipMemcpyAsync(G.o_rowptr, o_rowptr.data(), sizeof(int) * (G.n + 1), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_graph_scatter, g, b, 0, c->stream, G.n, G.rowptr, G.col, G.eid, G.rank, G.o_rowptr, G.o_col, G.o_eid);
    NRS_HIP(c, hipGetLastError());
    const int tot = o_rowptr[G.n];
    o_col.resize(tot);
    o_eid.resize(tot);
    if (tot) {
        NRS_HIP(c, hipMemcpyAsync(o_col.data(), G.o_col, sizeof(int) * tot, hipMemcpyDeviceToHost, c->stream));
        NRS_HIP(c, hipMemcpyAsync(o_eid.data(), G.o_eid, sizeof(int) * tot, hipMemcpyDeviceToHost, c->stream));
    }
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    return NRS_OK;
}

static int graph_update(nrs_ctx* c, GraphDevice& G, nrs_graph* g, const float* pos, int n_ids, const int* ids,
                        int* good) {
    if (n_ids == 0) return NRS_OK;
    NRS_HIP(c, hipMemcpyAsync(G.pos, pos, sizeof(float) * 3 * (size_t)G.n, hipMemcpyHostToDevice, c->stream));
    NRS_HIP(c, hipMemcpyAsync(G.ids, ids, sizeof(int) * n_ids, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_graph_update, dim3((n_ids + 255) / 256), dim3(256), 0, c->stream, n_ids, G.ids, G.rowptr, G.col,
                       G.eid, G.pos, G.w, G.mx, G.mn, G.status, g->sigma, g->stretch_th, G.good);
    NRS_HIP(c, hipGetLastError());
    NRS_HIP(c, hipMemcpyAsync(good, G.good, sizeof(int) * n_ids, hipMemcpyDeviceToHost, c->stream));
    if (G.ne) {
        NRS_HIP(c, hipMemcpyAsync(g->e_w, G.w, sizeof(float) * G.ne, hipMemcpyDeviceToHost, c->stream));
        NRS_HIP(c, hipMemcpyAsync(g->e_max, G.mx, sizeof(float) * G.ne, hipMemcpyDeviceToHost, c->stream));
        NRS_HIP(c, hipMemcpyAsync(g->e_min, G.mn, sizeof(float) * G.ne, hipMemcpyDeviceToHost, c->stream));
        NRS_HIP(c, hipMemcpyAsync(g->e_status, G.status, sizeof(int) * G.ne, hipMemcpyDeviceToHost, c->stream));
    }
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    return NRS_OK;
}

// Where a2 gets RegularizationGraph::GetEdges / UpdateVertex from: the host-owned flat graph (nrs_graph) or the
// device-resident dense one (nrs_rgraph: the reference's all-pairs density).
struct NeighbourSource {
    int n_points = 0;
    virtual ~NeighbourSource() {}
    // GetEdges of the map points in `want` (a source may serve every point), in the reference's order: entries beg[p] .. end[p] of
    // (col = other, w = weight, d0 = first distance, st = status) for a map point p that was asked for; the arrays are the source's
    // (valid until its next select)
    virtual int select(const std::vector<int>& want) = 0;
    std::vector<int> beg, end;
    const int* col = nullptr; const float* w = nullptr; const float* d0 = nullptr; const int* st = nullptr;
    // UpdateVertex of the listed points from the last world positions; good[i] = its return value
    virtual int update(const float* map_pos, int n, const int* ids, int* good) = 0;
    std::vector<char> truncated;      // per point: select() returned only a prefix of its list
    // per map point, may be null: 1 = the caller's walk passes over this point's connections without any effect unless they are
    // BAD; a source may leave them out of the lists (the dense one does: an embedded-mode walk would read ~N/M entries per node found)
    const std::vector<uint8_t>* pass_over = nullptr;
    virtual bool grow() { return false; }                          // fetch longer prefixes next time (false: there is nothing longer)
    // GetEdges of `want` AND the walk of OPT:252-279 over the lists where they are (a source that holds them on the device): per
    // wanted point (in order) the connections its walk accepts -- indices among `want` -- with weight and first distance, whether the
    // walk ended before its list did, and the lost-point flags.  *done = false: not available (the caller selects and walks itself).
    struct WalkOut { std::vector<int> n_acc, acc; std::vector<float> w, d0; std::vector<uint8_t> ended, lost; int passes = 0; };
    virtual int device_walk(const std::vector<int>&, const std::vector<int>&, const uint8_t*, WalkOut&, bool* done) { *done = false; return NRS_OK; }
    virtual void prefix_hint(int) {}                               // the next walks read about this many entries (grow() still applies)
};

struct FlatSource : NeighbourSource {
    nrs_ctx* c; nrs_graph* g; GraphDevice G;
    int init() { n_points = g->n_points; return graph_upload(c, G, g); }
    std::vector<int> rp_, col_, st_;
    std::vector<float> w_, d0_;
    int select(const std::vector<int>&) override {
        std::vector<int> eid;
        NRS_TRY(graph_select(c, G, g->sigma, rp_, col_, eid));
        w_.resize(eid.size()); d0_.resize(eid.size()); st_.resize(eid.size());
        for (size_t a = 0; a < eid.size(); ++a) { w_[a] = g->e_w[eid[a]]; d0_[a] = g->e_d0[eid[a]]; st_[a] = g->e_status[eid[a]]; }
        beg.assign(rp_.begin(), rp_.end() - 1); end.assign(rp_.begin() + 1, rp_.end());
        col = col_.data(); w = w_.data(); d0 = d0_.data(); st = st_.data();
        return NRS_OK;
    }
    int update(const float* map_pos, int n, const int* ids, int* good) override { return graph_update(c, G, g, map_pos, n, ids, good); }
};

int rg_get_edges_staged(nrs_rgraph* g, int32_t n_ids, const int32_t* ids, int32_t cap_per_point, const int** count, const int** col,
                        const int** status, const float** w, const float** d0, const uint8_t* pass_over, bool lists_to_host = true);   // nrs_rgraph.hip
int rg_walk(nrs_rgraph* g, int n_map, const int* code, const uint8_t* is_node, int* n_acc, int* acc, float* acc_w, float* acc_d0,
            uint8_t* ended, uint8_t* lost, int* converged, int* passes);
int rg_capacity(const nrs_rgraph* g);
int rg_max_cap_per_point(const nrs_rgraph* g);

struct DenseSource : NeighbourSource {
    nrs_ctx* c; nrs_rgraph* g; int cap;
    int select(const std::vector<int>& want) override {           // (the lists are read where they land: the graph's pinned staging area)
        const size_t n = (size_t)n_points, m = want.size();
        beg.assign(n, 0); end.assign(n, 0);
        truncated.assign(n, 0);
        col = nullptr; w = nullptr; d0 = nullptr; st = nullptr;
        if (m == 0) return NRS_OK;
        const int* cnt;
        NRS_TRY(rg_get_edges_staged(g, (int32_t)m, want.data(), cap, &cnt, &col, &st, &w, &d0, pass_over ? pass_over->data() : nullptr));
        for (size_t r = 0; r < m; ++r) {
            const int p = want[r];
            truncated[p] = cnt[r] > cap;
            beg[p] = (int)(r * (size_t)cap); end[p] = beg[p] + std::min(cnt[r], cap);
        }
        return NRS_OK;
    }
    int device_walk(const std::vector<int>& want, const std::vector<int>& code, const uint8_t* is_node, WalkOut& o, bool* done) override {
        *done = false;
        const size_t n = (size_t)n_points, m = want.size();
        if (m == 0 || code.size() != n) return NRS_OK;
        const int *cnt, *c1, *c2; const float *f1, *f2;
        NRS_TRY(rg_get_edges_staged(g, (int32_t)m, want.data(), cap, &cnt, &c1, &c2, &f1, &f2, pass_over ? pass_over->data() : nullptr, false));
        truncated.assign(n, 0);
        for (size_t r = 0; r < m; ++r) truncated[want[r]] = cnt[r] > cap;
        o.n_acc.resize(m); o.acc.resize(11 * m); o.w.resize(11 * m); o.d0.resize(11 * m); o.ended.resize(m); o.lost.resize(n);
        int conv = 0;
        NRS_TRY(rg_walk(g, (int)n, code.data(), is_node, o.n_acc.data(), o.acc.data(), o.w.data(), o.d0.data(), o.ended.data(), o.lost.data(), &conv, &o.passes));
        *done = conv != 0;
        return NRS_OK;
    }
    // (longer prefixes up to what one row's sort buffer holds in LDS; a walk that needs more than that is reported, not cut)
    bool grow() override {
        const int lim = std::min(n_points, rg_max_cap_per_point(g));
        if (cap >= lim) return false;
        cap = std::min(lim, 4 * cap);
        return true;
    }
    void prefix_hint(int n) override { cap = std::min(cap, std::max(n, 16)); }
    int update(const float* map_pos, int n, const int* ids, int* good) override { return n ? nrs_rgraph_update(g, map_pos, n, ids, good) : NRS_OK; }
};

static int track_core(nrs_ctx* c, const nrs_camera* cam, NeighbourSource& src, float* map_pos, int32_t n_f, const int32_t* f_map,
                      int32_t* f_status, const float* f_uv, float* f_pos, double pose_qt[7], float scale, float* deform_median,
                      int32_t* n_lost, int32_t* lost, nrs_lm_trace* trace, const uint8_t* f_node = nullptr);

}  // namespace nrs

using namespace nrs;

extern "C" int nrs_graph_select_neighbours(nrs_ctx* c, const nrs_graph* g, int32_t* o_rowptr, int32_t* o_col,
                                           int32_t* o_eid) {
    if (!c) return NRS_ERR_INVALID;
    NRS_TRY(graph_validate(c, g));
    if (!o_rowptr || (g->rowptr[g->n_points] > 0 && (!o_col || !o_eid))) return c->fail(NRS_ERR_INVALID, "null output");
    NRS_HIP(c, hipSetDevice(c->device));
    GraphDevice G;
    NRS_TRY(graph_upload(c, G, g));
    std::vector<int> rp, oc, oe;
    NRS_TRY(graph_select(c, G, g->sigma, rp, oc, oe));
    std::copy(rp.begin(), rp.end(), o_rowptr);
    std::copy(oc.begin(), oc.end(), o_col);
    std::copy(oe.begin(), oe.end(), o_eid);
    return NRS_OK;
}

extern "C" int nrs_graph_update(nrs_ctx* c, nrs_graph* g, const float* pos, int32_t n_ids, const int32_t* ids,
                                int32_t* good_count) {
    if (!c) return NRS_ERR_INVALID;
    NRS_TRY(graph_validate(c, g));
    if (n_ids < 0 || (n_ids > 0 && (!ids || !good_count || !pos))) return c->fail(NRS_ERR_INVALID, "nrs_graph_update: bad argument");
    for (int i = 0; i < n_ids; ++i)
        if (ids[i] < 0 || ids[i] >= g->n_points) return c->fail(NRS_ERR_INVALID, "point index out of range");
    NRS_HIP(c, hipSetDevice(c->device));
    GraphDevice G;
    NRS_TRY(graph_upload(c, G, g));
    return graph_update(c, G, g, pos, n_ids, ids, good_count);
}

extern "C" int nrs_track_deform_solve(nrs_ctx* c, const nrs_camera* cam, nrs_graph* g, float* map_pos,
                                      int32_t n_f, const int32_t* f_map, int32_t* f_status, const float* f_uv,
                                      float* f_pos, double pose_qt[7], float scale, float* deform_median,
                                      int32_t* n_lost, int32_t* lost, nrs_lm_trace* trace) {
    if (!c) return NRS_ERR_INVALID;
    if (!cam || !map_pos || n_f < 0 || !pose_qt || !n_lost || (n_f > 0 && (!f_map || !f_status || !f_uv || !f_pos)))
        return c->fail(NRS_ERR_INVALID, "nrs_track_deform_solve: bad argument");
    if (cam->model != NRS_CAM_PINHOLE && cam->model != NRS_CAM_KB8) return c->fail(NRS_ERR_INVALID, "unknown camera model %d", cam->model);
    NRS_TRY(graph_validate(c, g));
    NRS_HIP(c, hipSetDevice(c->device));
    FlatSource src;
    src.c = c; src.g = g;
    NRS_TRY(src.init());
    return track_core(c, cam, src, map_pos, n_f, f_map, f_status, f_uv, f_pos, pose_qt, scale, deform_median, n_lost, lost, trace);
}

// The same function on the device-resident dense graph (include/nrs.h): GetEdges / UpdateVertex see all N - 1
// connections of a point, as in the reference.
extern "C" int nrs_track_deform_solve_rg(nrs_ctx* c, const nrs_camera* cam, nrs_rgraph* g, int32_t n_points, int32_t cap_per_point,
                                         float* map_pos, int32_t n_f, const int32_t* f_map, int32_t* f_status, const float* f_uv,
                                         float* f_pos, double pose_qt[7], float scale, float* deform_median, int32_t* n_lost,
                                         int32_t* lost, nrs_lm_trace* trace) {
    if (!c) return NRS_ERR_INVALID;
    if (!cam || !g || !map_pos || n_points <= 0 || cap_per_point <= 0 || n_f < 0 || !pose_qt || !n_lost || (n_f > 0 && (!f_map || !f_status || !f_uv || !f_pos)))
        return c->fail(NRS_ERR_INVALID, "nrs_track_deform_solve_rg: bad argument");
    if (cam->model != NRS_CAM_PINHOLE && cam->model != NRS_CAM_KB8) return c->fail(NRS_ERR_INVALID, "unknown camera model %d", cam->model);
    // the dense state is capacity x capacity and map_pos has one row per point of it: a different n_points would index either
    // past the end (regularization_graph.cc has no such failure mode: its maps are keyed by ID)
    if (n_points != rg_capacity(g))
        return c->fail(NRS_ERR_INVALID, "nrs_track_deform_solve_rg: n_points %d is not the graph's capacity %d", n_points, rg_capacity(g));
    NRS_HIP(c, hipSetDevice(c->device));
    DenseSource src;
    src.c = c; src.g = g; src.cap = std::min(cap_per_point, rg_max_cap_per_point(g)); src.n_points = n_points;
    return track_core(c, cam, src, map_pos, n_f, f_map, f_status, f_uv, f_pos, pose_qt, scale, deform_median, n_lost, lost, trace);
}

// N2 (include/nrs.h): the embedded-deformation mode on the device-resident dense graph
extern "C" int nrs_track_deform_solve_embedded(nrs_ctx* c, const nrs_camera* cam, nrs_rgraph* g, int32_t n_points, int32_t cap_per_point,
                                               float* map_pos, int32_t n_f, const int32_t* f_map, int32_t* f_status, const float* f_uv,
                                               float* f_pos, const uint8_t* f_node, double pose_qt[7], float scale, float* deform_median,
                                               int32_t* n_lost, int32_t* lost, nrs_lm_trace* trace) {
    if (!c) return NRS_ERR_INVALID;
    if (!cam || !g || !map_pos || n_points <= 0 || cap_per_point <= 0 || n_f < 0 || !pose_qt || !n_lost || (n_f > 0 && (!f_map || !f_status || !f_uv || !f_pos || !f_node)))
        return c->fail(NRS_ERR_INVALID, "nrs_track_deform_solve_embedded: bad argument");
    if (cam->model != NRS_CAM_PINHOLE && cam->model != NRS_CAM_KB8) return c->fail(NRS_ERR_INVALID, "unknown camera model %d", cam->model);
    if (n_points != rg_capacity(g))
        return c->fail(NRS_ERR_INVALID, "nrs_track_deform_solve_embedded: n_points %d is not the graph's capacity %d", n_points, rg_capacity(g));
    NRS_HIP(c, hipSetDevice(c->device));
    DenseSource src;
    src.c = c; src.g = g; src.cap = std::min(cap_per_point, rg_max_cap_per_point(g)); src.n_points = n_points;
    return track_core(c, cam, src, map_pos, n_f, f_map, f_status, f_uv, f_pos, pose_qt, scale, deform_median, n_lost, lost, trace, f_node);
}

namespace nrs {
// f_node (may be null: every optimised point is a node = the reference function): the EMBEDDED-DEFORMATION mode (N2, SURVEY.md 8d;
// stated in oracle/embedded_oracle.py).  Nodes carry the vertices and the regularisers of OPT:255-335; every other optimised point is
// skinned to the <= 11 nodes its own GetEdges walk accepts (normalised connection weights) and its reprojection edge constrains them.
static int track_core(nrs_ctx* c, const nrs_camera* cam, NeighbourSource& src, float* map_pos, int32_t n_f, const int32_t* f_map,
                      int32_t* f_status, const float* f_uv, float* f_pos, double pose_qt[7], float scale, float* deform_median,
                      int32_t* n_lost, int32_t* lost, nrs_lm_trace* trace, const uint8_t* f_node) {
    const bool tm = c->env("NRS_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!tm) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[nrs] a2 %-22s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    if (trace) { trace->count = 0; trace->iterations = 0; }
    *n_lost = 0;
    if (deform_median) *deform_median = 0.f;
    const int n_map = src.n_points;
    std::vector<int> map_to_frame(n_map, -1);
    for (int i = 0; i < n_f; ++i) {
        if (f_map[i] >= n_map) return c->fail(NRS_ERR_INVALID, "f_map out of range");
        if (f_map[i] >= 0) map_to_frame[f_map[i]] = i;
    }
    // points in the optimisation: TRACKED_WITH_3D in frame index order (OPT:174-192); the nodes among them carry the vertices
    std::vector<int> opt_f, ids;
    for (int i = 0; i < n_f; ++i)
        if (f_status[i] == NRS_TRACKED_WITH_3D && f_map[i] >= 0) { opt_f.push_back(i); ids.push_back(f_map[i]); }
    const int N = (int)opt_f.size();
    if (N == 0) return NRS_OK;                       // nothing to optimise (g2o: empty graph)
    std::vector<int> id_to_idx(n_map, -1), node_of(N, -1), node_idx;
    for (int i = 0; i < N; ++i) {
        id_to_idx[ids[i]] = i;
        if (!f_node || f_node[opt_f[i]]) { node_of[i] = (int)node_idx.size(); node_idx.push_back(i); }
    }
    const int M = (int)node_idx.size();
    if (M == 0) return c->fail(NRS_ERR_INVALID, "embedded mode: no node among the optimised points");

    // ---- edge construction OPT:224-337 (container walk on the host, order as in the reference)
    // The reference keeps per vertex the (other, edge) pairs it is part of and skips a neighbour it is already paired with (OPT:268-272).
    // A pair {idx, io} exists when idx's walk reaches io iff io was walked EARLIER (io < idx) and accepted idx (a list holds a connection
    // once): the test reads io's accepted neighbours -- at most 11, one cache line -- instead of a container per vertex
    std::vector<int> acc(11 * (size_t)N, -1);
    std::vector<uint8_t> n_acc(N, 0);
    std::vector<int> dm_idx, sp_ij;
    std::vector<float> dm_w, sp_d0;
    dm_idx.reserve(48 * (size_t)M); sp_ij.reserve(24 * (size_t)M); dm_w.reserve(12 * (size_t)M); sp_d0.reserve(12 * (size_t)M);
    std::vector<int> sk_node((size_t)(N - M) * 11, -1), sk_of(N, -1), sk_idx;     // skinned observations: nodes (vertex indices), weights
    std::vector<double> sk_om((size_t)(N - M) * 11, 0.0);
    std::vector<uint8_t> lost_flag(n_map, 0);                     // btree_set<ID> (OPT:222) as a flag per id: read out in ascending order below
    std::vector<uint8_t> no_vertex;                               // embedded mode: optimised points without a vertex (passed over below)
    if (M < N) {
        no_vertex.assign(n_map, 0);
        for (int i = 0; i < N; ++i) no_vertex[ids[i]] = node_of[i] < 0;
        src.pass_over = &no_vertex;
    }
    // what the walk does with a connection to map point o (OPT:262-275): >= 0 its index among the optimised points (a vertex: an edge, or
    // a skinning weight), -1 nothing (not in the frame, just triangulated, or optimised without a vertex: passed over), -2 a lost point
    std::vector<int> walk_code(n_map, -1);
    for (int o = 0; o < n_map; ++o) {
        const int fo = map_to_frame[o];
        if (fo < 0) continue;
        if (f_status[fo] != NRS_TRACKED_WITH_3D) { if (f_status[fo] != NRS_JUST_TRIANGULATED) walk_code[o] = -2; continue; }
        const int io = id_to_idx[o];
        if (io >= 0 && node_of[io] >= 0) walk_code[o] = io;
    }
    const bool host_walk = c->env("NRS_HOST_WALK") != nullptr;     // (A/B switch: the walk on the host, as before round 5)
    std::vector<uint8_t> is_node_b;
    if (M < N) { is_node_b.resize(N); for (int i = 0; i < N; ++i) is_node_b[i] = node_of[i] >= 0; }
    for (bool again = true; again;) {                             // (again: a walk ran off a truncated list -- longer prefixes, from the start)
    again = false;
    if (!host_walk) {
        // the dense graph walks on the device (nrs_rgraph.hip k_rg_walk): the lists never leave it, what comes back are the <= 11 accepted
        // connections per point; the edges are made from them here in the order the sequential walk makes them
        NeighbourSource::WalkOut wo;
        bool dev = false;
        const auto tw0 = std::chrono::steady_clock::now();
        NRS_TRY(src.device_walk(ids, walk_code, M < N ? is_node_b.data() : nullptr, wo, &dev));
        if (tm) fprintf(stderr, "[nrs] a2 device_walk call %.2f ms (N %d)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count(), N);
        if (tm) fprintf(stderr, "[nrs] a2 device walk: %d passes, %s\n", wo.passes, dev ? "converged" : "not taken");
        if (dev) {
            mark("GetEdges + device walk");
            size_t n_e = 0;                                        // (the edge arrays are sized once and written by index: ~50 ns an edge with four vector appends)
            for (int idx = 0; idx < N; ++idx) if (node_of[idx] >= 0) n_e += wo.n_acc[idx];
            dm_idx.resize(4 * n_e); sp_ij.resize(2 * n_e); dm_w.resize(n_e); sp_d0.resize(n_e);
            size_t ne = 0;
            std::copy(wo.lost.begin(), wo.lost.end(), lost_flag.begin());
            sk_idx.clear(); std::fill(sk_of.begin(), sk_of.end(), -1); std::fill(sk_node.begin(), sk_node.end(), -1); std::fill(sk_om.begin(), sk_om.end(), 0.0);
            for (int idx = 0; idx < N && !again; ++idx) {
                const bool is_node = node_of[idx] >= 0;
                const size_t slot = sk_idx.size();
                const int nr = wo.n_acc[idx];
                double wsum = 0;
                for (int k = 0; k < nr; ++k) {
                    const int io = wo.acc[11 * (size_t)idx + k];
                    const float wk = wo.w[11 * (size_t)idx + k];
                    if (is_node) {
                        const int a = node_of[idx], b = node_of[io];
                        int* dq = &dm_idx[4 * ne];
                        dq[0] = -1; dq[1] = -1; dq[2] = a; dq[3] = b;        // r = w (delta_idx - delta_io)
                        dm_w[ne] = wk;
                        sp_ij[2 * ne] = a; sp_ij[2 * ne + 1] = b;
                        sp_d0[ne] = wo.d0[11 * (size_t)idx + k];
                        ++ne;
                    } else {
                        sk_node[11 * slot + k] = node_of[io];
                        sk_om[11 * slot + k] = (double)wk;
                        wsum += (double)wk;
                    }
                }
                if (!is_node && nr > 0) {
                    for (int k = 0; k < nr; ++k) sk_om[11 * slot + k] /= wsum;
                    sk_of[idx] = (int)slot;
                    sk_idx.push_back(idx);
                }
                if (!wo.ended[idx] && !src.truncated.empty() && src.truncated[ids[idx]]) {
                    if (!src.grow()) return c->fail(NRS_ERR_INVALID, "the neighbour walk of map point %d ran off its list", ids[idx]);
                    again = true;
                    if (tm) fprintf(stderr, "[nrs] a2 device walk: point %d ran off its list: longer prefixes\n", idx);
                }
            }
            continue;
        }
    }
    NRS_TRY(src.select(ids));                                      // the walks below start from the optimised points only
    const int *ocol = src.col, *ost = src.st;
    const float *ow = src.w, *od0 = src.d0;
    mark("GetEdges");
    std::fill(n_acc.begin(), n_acc.end(), 0);
    size_t ne = 0;                                                 // (at most 11 edges a walk: sized for that, written by index, cut to size below)
    dm_idx.resize(44 * (size_t)M); sp_ij.resize(22 * (size_t)M); dm_w.resize(11 * (size_t)M); sp_d0.resize(11 * (size_t)M);
    std::fill(lost_flag.begin(), lost_flag.end(), 0);
    sk_idx.clear(); std::fill(sk_of.begin(), sk_of.end(), -1); std::fill(sk_node.begin(), sk_node.end(), -1); std::fill(sk_om.begin(), sk_om.end(), 0.0);
    for (int idx = 0; idx < N && !again; ++idx) {
        const int p = ids[idx];
        if (idx + 6 < N) {                                        // (the lists have just arrived from the device: every walk would start on lines that are in no cache)
            const int pb = src.beg[ids[idx + 6]];
            for (int o = 0; o < 32; o += 16) {
                __builtin_prefetch(ocol + pb + o); __builtin_prefetch(ost + pb + o);
                __builtin_prefetch(ow + pb + o); __builtin_prefetch(od0 + pb + o);
            }
        }
        const bool is_node = node_of[idx] >= 0;
        const size_t slot = sk_idx.size();                        // (a skinned observation's slot, kept only if it meets a node)
        int n_reg = 0;
        double wsum = 0;
        bool ended = false;
        for (int a = src.beg[p]; a < src.end[p]; ++a) {
            const int other = ocol[a];
            if (n_reg > 10 || ost[a] == NRS_GRAPH_BAD) { ended = true; break; }
            const int io = walk_code[other];                      // (one look-up instead of four dependent ones: ~2 x 10^5 entries are walked at 4.4k points)
            if (io < 0) {
                if (io == -2) lost_flag[other] = 1;
                continue;
            }
            if (is_node) {
                bool dup = false;
                if (io < idx) { const int* al = &acc[11 * (size_t)io]; for (int k = 0, nk = n_acc[io]; k < nk; ++k) dup = dup || al[k] == idx; }
                if (dup) continue;
                const int va = node_of[idx], vb = node_of[io];
                int* dq = &dm_idx[4 * ne];
                dq[0] = -1; dq[1] = -1; dq[2] = va; dq[3] = vb;       // r = w (delta_idx - delta_io)
                dm_w[ne] = ow[a];
                sp_ij[2 * ne] = va; sp_ij[2 * ne + 1] = vb;
                sp_d0[ne] = od0[a];
                ++ne;
                acc[11 * (size_t)idx + n_acc[idx]++] = io;           // (n_reg <= 10 here: at most 11 per walk)
            } else {
                sk_node[11 * slot + n_reg] = node_of[io];
                sk_om[11 * slot + n_reg] = (double)ow[a];
                wsum += (double)ow[a];
            }
            ++n_reg;
        }
        if (!is_node && n_reg > 0) {                              // omega = w / sum w (float weights, double arithmetic)
            for (int k = 0; k < n_reg; ++k) sk_om[11 * slot + k] /= wsum;
            sk_of[idx] = (int)slot;
            sk_idx.push_back(idx);
        } else if (!is_node) {
            for (int k = 0; k < 11; ++k) { sk_node[11 * slot + k] = -1; sk_om[11 * slot + k] = 0.0; }
        }
        if (!ended && !src.truncated.empty() && src.truncated[p]) {
            if (!src.grow()) return c->fail(NRS_ERR_INVALID, "the neighbour walk of map point %d ran off its list", p);
            again = true;
        }
    }
    dm_idx.resize(4 * ne); sp_ij.resize(2 * ne); dm_w.resize(ne); sp_d0.resize(ne);
    }
    src.pass_over = nullptr;
    mark("GetEdges + edge construction");
    const int E = (int)dm_w.size(), S = (int)sk_idx.size();

    // ---- engine for the two inlier rounds: one vertex per node
    EngineSpec s;
    Pose seed;
    for (int i = 0; i < 4; ++i) seed.q[i] = pose_qt[i];
    for (int i = 0; i < 3; ++i) seed.t[i] = pose_qt[4 + i];
    quat_normalize(seed.q);
    std::vector<double> X0(3 * (size_t)M), zeros(3 * (size_t)M, 0.0), skX0(3 * (size_t)S);
    std::vector<float> uv(2 * (size_t)M), skuv(2 * (size_t)S);
    std::vector<int> lm_pose(M, 0);
    for (int v = 0; v < M; ++v) {
        const int fi = opt_f[node_idx[v]];
        for (int k = 0; k < 3; ++k) X0[3 * (size_t)v + k] = (double)f_pos[3 * (size_t)fi + k];
        uv[2 * (size_t)v] = f_uv[2 * (size_t)fi];
        uv[2 * (size_t)v + 1] = f_uv[2 * (size_t)fi + 1];
    }
    for (int q = 0; q < S; ++q) {
        const int fi = opt_f[sk_idx[q]];
        for (int k = 0; k < 3; ++k) skX0[3 * (size_t)q + k] = (double)f_pos[3 * (size_t)fi + k];
        skuv[2 * (size_t)q] = f_uv[2 * (size_t)fi];
        skuv[2 * (size_t)q + 1] = f_uv[2 * (size_t)fi + 1];
    }
    std::vector<uint8_t> rflag(M, RF_OBS | RF_REPROJ_ACTIVE), dm_active(E, 1), sk_active(S, 1);
    s.K = 1; s.M = M;
    s.poses = &seed;
    s.x = zeros.data(); s.X0 = X0.data();
    s.lm_pose = lm_pose.data(); s.uv = uv.data(); s.rflag = rflag.data();
    s.n_sp = E; s.sp_ij = sp_ij.data(); s.sp_d0 = sp_d0.data();
    s.n_dm = E; s.dm_idx = dm_idx.data(); s.dm_w = dm_w.data(); s.dm_active = dm_active.data();
    s.n_skin = S; s.sk_uv = skuv.data(); s.sk_X0 = skX0.data(); s.sk_node = sk_node.data(); s.sk_om = sk_om.data();
    s.cam.model = cam->model;
    for (int i = 0; i < 8; ++i) s.cam.p[i] = cam->params[i];
    ba_constants(s, scale);
    s.delta_pos = s.delta_spatial;                                // Huber sqrt(0.584) on the springs (OPT:324-326)
    s.spring_form = 1;
    Engine* eng = nullptr;
    NRS_TRY(engine_create(c, s, &c->arena_trk, &eng));
    struct EG { nrs_ctx* c; Engine* e; ~EG() { engine_destroy(c, e); } } eg{c, eng};

    mark("engine 1");
    const float th2_sq = 5.99f, th3_sq = 0.584f;
    std::vector<char> inl(N, 1);
    std::vector<double> chi_r(M), chi_d(E), chi_s(S);
    for (int rnd = 0; rnd < 2; ++rnd) {                          // OPT:338-395
        NRS_TRY(engine_reset(c, eng));
        NRS_TRY(engine_optimize(c, eng, 10, rnd, trace));
        mark("  round: optimize");
        NRS_TRY(engine_edge_chi2(c, eng, chi_r.data(), nullptr, chi_d.data()));
        mark("  round: edge chi2");
        for (int v = 0; v < M; ++v) {
            const int idx = node_idx[v];
            const bool out = (float)chi_r[v] > th2_sq;
            inl[idx] = !out;
            rflag[v] = RF_OBS | (out ? 0 : RF_REPROJ_ACTIVE);
        }
        // OPT:365-383 sets the level of every regulariser of a vertex twice -- by the vertex's reprojection gate, then by the edge's own
        // chi2 -- and the second assignment stands: every edge (each has a vertex) ends at its own gate
        for (int k = 0; k < E; ++k) dm_active[k] = chi_d[k] > (double)th3_sq ? 0 : 1;
        // (the levels decide what the NEXT round optimises: after the last round only their host copies are read -- by stage 2 -- and this engine is not optimised again)
        if (rnd < 1) NRS_TRY(engine_update_flags(c, eng, rflag.data(), nullptr, nullptr, dm_active.data()));
        mark("  round: levels");
        if (S) {                                                  // the skinned observations' levels, by the same gate
            NRS_TRY(engine_skin_chi2(c, eng, chi_s.data()));
            for (int q = 0; q < S; ++q) { const bool out = (float)chi_s[q] > th2_sq; inl[sk_idx[q]] = !out; sk_active[q] = out ? 0 : 1; }
            if (rnd < 1) NRS_TRY(engine_skin_set_active(c, eng, sk_active.data()));
        }
    }
    mark("two rounds");
    Pose pose_out;
    std::vector<double> delta_v(3 * (size_t)M), delta(3 * (size_t)N, 0.0);
    NRS_TRY(engine_download(c, eng, &pose_out, delta_v.data()));
    for (int i = 0; i < 4; ++i) pose_qt[i] = pose_out.q[i];
    for (int i = 0; i < 3; ++i) pose_qt[4 + i] = pose_out.t[i];
    for (int v = 0; v < M; ++v)
        for (int k = 0; k < 3; ++k) delta[3 * (size_t)node_idx[v] + k] = delta_v[3 * (size_t)v + k];
    for (int q = 0; q < S; ++q)                                   // a skinned point's deformation: the interpolated one
        for (int k = 0; k < 3; ++k) {
            double a = 0;
            for (int j = 0; j < 11; ++j)
                if (sk_node[11 * (size_t)q + j] >= 0) a += sk_om[11 * (size_t)q + j] * delta_v[3 * (size_t)sk_node[11 * (size_t)q + j] + k];
            delta[3 * (size_t)sk_idx[q] + k] = a;
        }

    // ---- OPT:401-455: deformation statistics, status / position updates (all optimised points alike)
    std::vector<float> mag(N), dfl(3 * (size_t)N);
    for (int i = 0; i < N; ++i) {
        const float d0 = (float)delta[3 * (size_t)i], d1 = (float)delta[3 * (size_t)i + 1], d2 = (float)delta[3 * (size_t)i + 2];
        dfl[3 * (size_t)i] = d0; dfl[3 * (size_t)i + 1] = d1; dfl[3 * (size_t)i + 2] = d2;
        mag[i] = std::sqrt(d0 * d0 + d1 * d1 + d2 * d2);
    }
    std::vector<float> srt = mag;                                 // (the two order statistics of the sorted magnitudes, without sorting all of them)
    const int i1 = (int)(N * 0.25f), i3 = (int)(N * 0.75f);
    std::nth_element(srt.begin(), srt.begin() + i3, srt.end());
    std::nth_element(srt.begin(), srt.begin() + i1, srt.begin() + i3);
    const float q1 = srt[i1], q3 = srt[i3];
    const float th = 1.5f * (q3 - q1);
    for (int idx = 0; idx < N; ++idx) {
        const int fi = opt_f[idx];
        const double chi = node_of[idx] >= 0 ? chi_r[node_of[idx]] : (sk_of[idx] >= 0 ? chi_s[sk_of[idx]] : 0.0);
        if ((float)chi > th2_sq) { inl[idx] = 0; f_status[fi] = NRS_TRACKED; }
        if (mag[idx] >= q3 + th) { f_status[fi] = NRS_TRACKED; continue; }
        if (node_of[idx] >= 0) rflag[node_of[idx]] |= RF_FIXED;
        for (int k = 0; k < 3; ++k) {
            const float cur = dfl[3 * (size_t)idx + k] + f_pos[3 * (size_t)fi + k];
            f_pos[3 * (size_t)fi + k] = cur;
            map_pos[3 * (size_t)ids[idx] + k] = cur;
        }
    }
    if (deform_median) {
        std::vector<float> m2 = mag;
        std::nth_element(m2.begin(), m2.begin() + N / 2, m2.end());
        *deform_median = m2[N / 2];
    }
    mark("statistics");
    // ---- graph update OPT:457-474
    {
        std::vector<int> upd_ids, upd_idx;
        for (int idx = 0; idx < N; ++idx)
            if (inl[idx]) { upd_ids.push_back(ids[idx]); upd_idx.push_back(idx); }
        std::vector<int> good(upd_ids.size());
        NRS_TRY(src.update(map_pos, (int)upd_ids.size(), upd_ids.data(), good.data()));
        for (size_t i = 0; i < upd_ids.size(); ++i)
            if (good[i] < 10 * 0.5) f_status[opt_f[upd_idx[i]]] = NRS_BAD;
    }
    mark("UpdateVertex");
    std::vector<int> lost_ids;
    for (int i = 0; i < n_map; ++i) if (lost_flag[i]) lost_ids.push_back(i);
    if (lost_ids.empty()) return NRS_OK;

    // ---- stage 2 OPT:476-553: lost points follow their (fixed) optimised neighbours.  Vertices: the nodes, then the optimised points
    // without a vertex as constants (their interpolated deformation), then the lost points
    const int L = (int)lost_ids.size();
    std::vector<int> vert_of(N, -1), others;
    for (int v = 0; v < M; ++v) vert_of[node_idx[v]] = v;
    for (int idx = 0; idx < N; ++idx)
        if (node_of[idx] < 0) { vert_of[idx] = M + (int)others.size(); others.push_back(idx); }
    const int NV = M + (int)others.size();
    std::vector<int> un_ij;
    std::vector<float> un_w;
    // the walk below counts optimised neighbours only and stops after 11: everything else may stay out of the lists
    std::vector<uint8_t> not_optimised(n_map);
    for (int i = 0; i < n_map; ++i) not_optimised[i] = id_to_idx[i] < 0;
    src.pass_over = &not_optimised;
    src.prefix_hint(32);
    for (bool again = true; again;) {
    again = false;
    NRS_TRY(src.select(lost_ids));                                 // GetEdges sees the updated graph
    const int* ocol = src.col;
    const float* ow = src.w;
    un_ij.clear(); un_w.clear();
    for (int li = 0; li < L && !again; ++li) {
        const int p = lost_ids[li];
        if (li + 6 < L) { const int pb = src.beg[lost_ids[li + 6]]; __builtin_prefetch(ocol + pb); __builtin_prefetch(ow + pb); __builtin_prefetch(ocol + pb + 16); __builtin_prefetch(ow + pb + 16); }
        int n_reg = 0;
        bool ended = false;
        for (int a = src.beg[p]; a < src.end[p]; ++a) {
            if (n_reg > 10) { ended = true; break; }
            const int io = id_to_idx[ocol[a]];
            if (io < 0) continue;
            un_ij.insert(un_ij.end(), {NV + li, vert_of[io]});
            un_w.push_back(ow[a]);
            ++n_reg;
        }
        if (!ended && !src.truncated.empty() && src.truncated[p]) {
            if (!src.grow()) return c->fail(NRS_ERR_INVALID, "the neighbour walk of lost map point %d ran off its list", p);
            again = true;
        }
    }
    }
    src.pass_over = nullptr;
    mark("GetEdges 2 + walk");
    const int M2 = NV + L;
    // Only the free vertices (nodes the statistics left free, lost points) and what an edge ties them to take part: an edge between
    // two fixed vertices is not in the problem (g2o skips allVerticesFixed edges; the engine masks them) and a fixed vertex no
    // kept edge touches is read by nothing.  The engine is built on that part -- a few hundred vertices instead of all of them.
    std::vector<uint8_t> rflag_all(M2, 0);
    std::copy(rflag.begin(), rflag.end(), rflag_all.begin());
    for (size_t o = 0; o < others.size(); ++o) rflag_all[M + o] = RF_FIXED;
    std::vector<int> newid(M2, -1);
    std::vector<uint8_t> keep_v(M2, 0), keep_e(E, 0);
    for (int v = 0; v < M2; ++v) keep_v[v] = !(rflag_all[v] & RF_FIXED);
    for (int k = 0; k < E; ++k) {
        const int a = sp_ij[2 * (size_t)k], b = sp_ij[2 * (size_t)k + 1];
        if (!(rflag_all[a] & RF_FIXED) || !(rflag_all[b] & RF_FIXED)) keep_e[k] = 1;
    }
    for (int k = 0; k < E; ++k) if (keep_e[k]) { keep_v[sp_ij[2 * (size_t)k]] = 1; keep_v[sp_ij[2 * (size_t)k + 1]] = 1; }
    for (size_t q = 0; q < un_w.size(); ++q) { keep_v[un_ij[2 * q]] = 1; keep_v[un_ij[2 * q + 1]] = 1; }
    int M2k = 0;
    for (int v = 0; v < M2; ++v) if (keep_v[v]) newid[v] = M2k++;
    std::vector<double> x2(3 * (size_t)M2k, 0.0), X02(3 * (size_t)M2k, 0.0);
    std::vector<float> uv2(2 * (size_t)M2k, 0.f);
    std::vector<int> lm_pose2(M2k, 0);
    std::vector<uint8_t> rflag2(M2k, 0);
    for (int v = 0; v < M2; ++v) {
        const int nv = newid[v];
        if (nv < 0) continue;
        rflag2[nv] = rflag_all[v];
        if (v < M) {
            for (int k = 0; k < 3; ++k) { x2[3 * (size_t)nv + k] = delta_v[3 * (size_t)v + k]; X02[3 * (size_t)nv + k] = X0[3 * (size_t)v + k]; }
            uv2[2 * (size_t)nv] = uv[2 * (size_t)v]; uv2[2 * (size_t)nv + 1] = uv[2 * (size_t)v + 1];
        } else if (v < NV) {
            for (int k = 0; k < 3; ++k) x2[3 * (size_t)nv + k] = delta[3 * (size_t)others[v - M] + k];
        }
    }
    std::vector<int> sp_ij2, dm_idx2, un_ij2(un_ij.size());
    std::vector<float> sp_d02, dm_w2;
    std::vector<uint8_t> dm_active2;
    for (int k = 0; k < E; ++k) {
        if (!keep_e[k]) continue;
        const int a = newid[sp_ij[2 * (size_t)k]], b = newid[sp_ij[2 * (size_t)k + 1]];
        sp_ij2.insert(sp_ij2.end(), {a, b});
        sp_d02.push_back(sp_d0[k]);
        dm_idx2.insert(dm_idx2.end(), {-1, -1, newid[dm_idx[4 * (size_t)k + 2]], newid[dm_idx[4 * (size_t)k + 3]]});
        dm_w2.push_back(dm_w[k]);
        dm_active2.push_back(dm_active[k]);
    }
    for (size_t q = 0; q < un_ij.size(); ++q) un_ij2[q] = newid[un_ij[q]];
    const uint8_t pose_fixed = 1;
    EngineSpec s2 = s;
    s2.M = M2k;
    s2.poses = &pose_out;
    s2.pose_fixed = &pose_fixed;
    s2.x = x2.data(); s2.X0 = X02.data();
    s2.lm_pose = lm_pose2.data(); s2.uv = uv2.data(); s2.rflag = rflag2.data();
    s2.n_sp = (int)sp_d02.size(); s2.sp_ij = sp_ij2.data(); s2.sp_d0 = sp_d02.data();
    s2.n_dm = (int)dm_w2.size(); s2.dm_idx = dm_idx2.data(); s2.dm_w = dm_w2.data(); s2.dm_active = dm_active2.data();
    s2.n_un = (int)un_w.size(); s2.un_ij = un_ij2.data(); s2.un_w = un_w.data();
    s2.n_skin = 0;                                                // (the skinned observations take part in the two rounds only)
    engine_destroy(c, eng);
    eg.e = nullptr;
    Engine* eng2 = nullptr;
    NRS_TRY(engine_create(c, s2, &c->arena_trk, &eng2));
    eg.e = eng2;
    mark("engine 2");
    NRS_TRY(engine_optimize(c, eng2, 10, 2, trace));
    mark("stage 2 solve");
    std::vector<double> x_out(3 * (size_t)M2k);
    NRS_TRY(engine_download(c, eng2, nullptr, x_out.data()));
    for (int li = 0; li < L; ++li) {
        for (int k = 0; k < 3; ++k)
            map_pos[3 * (size_t)lost_ids[li] + k] = (float)x_out[3 * (size_t)newid[NV + li] + k] + map_pos[3 * (size_t)lost_ids[li] + k];
        if (lost) lost[li] = lost_ids[li];
    }
    *n_lost = L;
    return NRS_OK;
}
}  // namespace nrs
