// a19 / a20 at the reference's density: RegularizationGraph (reference modules/map/regularization_graph.{h,cc}) as a
// device-resident object.
//
// The reference graph is all-pairs: Map::InitializeRegularizationGraph adds an edge for EVERY pair of initial points
// (modules/map/map.cc:148-166) and Mapping adds every new landmark against every current one
// (modules/mapping/mapping.cc:240-256), so a vertex has N - 1 connections, GetEdges copies and sorts all of them per
// call (regularization_graph.cc:71-87) and UpdateVertex walks all of them (:130-146).  A radius cut-off is NOT
// equivalent: UpdateVertex's return value counts every connection that passes the stretch test, far ones included,
// and the caller marks a point BAD below 5 (g2o_optimization.cc:468-473).  So the graph is kept the way the hardware
// likes it -- dense: four capacity x capacity arrays (max / min / first distance as fp32, status as a byte), the pair
// (i, j) stored once at [min(i,j)][max(i,j)]; 5k points = 325 MB, 10k = 1.3 GB of the 288 GB.  The weight is not
// stored: it is InterpolationWeight(max_distance, sigma) at every point where the reference writes it (AddEdge :46,
// UpdateConnection :113), so it is re-formed where it is read.
//   add_edges : one thread per (new, other) pair                               O(1) per pair, coalesced rows
//   update    : one workgroup sweep per listed vertex over its row/column      O(deg), good_count by block reduction
//   get_edges : one wave per listed vertex: one pass finds the status class at which GetEdges' "break at the first
//               weight < min_weight" falls and compacts the surviving entries (they all lie within 1.5 sigma), which are
//               put in order as 64-bit keys (status | inverted weight bits | index): rank sort when they are a few hundred,
//               O(deg) + O(k^2 / 64), a bitonic sort in LDS beyond (a sigma that covers most of the map)
#include <algorithm>
#include <cmath>
#include <new>
#include <vector>
#include "nrs_ctx.hpp"

struct nrs_rgraph {
    nrs_ctx* c = nullptr;
    int cap = 0;
    float sigma = 1.f, stretch_th = 1.1f, min_w = 0.f;
    float *maxd = nullptr, *mind = nullptr, *d0 = nullptr;
    uint8_t* st = nullptr;
    nrs::DevBuf pos, ids_a, ids_b, out_i, out_f, good, skip, slot, walk;   // (walk: the device-side neighbour walk of a2, rg_walk)
    nrs::DevBuf mir;                 // status + longest distance once more as FULL matrices (k_rg_mirror), for GetEdges over many rows
    int last_n_ids = 0, last_cap = 0;  // shape of the lists out_i holds (the last GetEdges)
    std::vector<int> h_slot;
    char* pin = nullptr;             // pinned staging area of the GetEdges results (page-faulting pageable targets cost more than the kernel)
    size_t pin_cap = 0;
};

namespace nrs {

constexpr uint8_t RG_NONE = 0xFF;

// InterpolationWeight (utilities/geometry_toolbox.cc:26-28): float argument, exp evaluated in double and rounded
// to float (the definition shared with the oracle and nrs_track.hip, DESIGN.md "weights")
__host__ __device__ inline float rg_weight(float d, float sigma) {
#pragma clang fp contract(off)
    const float arg = -(d * d) / (2.0f * sigma * sigma);
    return (float)exp((double)arg);
}

__device__ inline size_t rg_at(int i, int j, int cap) { return i < j ? (size_t)i * cap + j : (size_t)j * cap + i; }

__device__ inline float rg_dist(const float* __restrict__ pos, int i, int j) {
#pragma clang fp contract(off)
    const float dx = pos[3 * i] - pos[3 * j], dy = pos[3 * i + 1] - pos[3 * j + 1], dz = pos[3 * i + 2] - pos[3 * j + 2];
    return sqrtf(dx * dx + dy * dy + dz * dz);
}

// AddEdge (regularization_graph.cc:38-55) for every (new, other) pair, new != other: a fresh NEUTRAL edge whose
// first / max / min distance is the current one.  A pair listed from both sides is written twice with the same bits.
__global__ void k_rg_add(int n_new, const int* __restrict__ new_ids, int n_other, const int* __restrict__ other_ids,
                         const float* __restrict__ pos, int cap, float* maxd, float* mind, float* d0, uint8_t* st) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x, a = blockIdx.y;
    if (b >= n_other || a >= n_new) return;
    const int i = new_ids[a], j = other_ids[b];
    if (i == j) return;
    const float d = rg_dist(pos, i, j);
    const size_t k = rg_at(i, j, cap);
    maxd[k] = d; mind[k] = d; d0[k] = d; st[k] = (uint8_t)NRS_GRAPH_NEUTRAL;
}

// UpdateVertex (regularization_graph.cc:130-146) + UpdateConnection (:89-128) for one listed vertex per blockIdx.y.
// Two listed vertices sharing an edge compute the same values from the same positions: idempotent.
__global__ __launch_bounds__(256) void k_rg_update(int n_ids, const int* __restrict__ ids, const float* __restrict__ pos, int cap,
                                                   float* maxd, float* mind, uint8_t* st, float stretch_th, int* good) {
#pragma clang fp contract(off)
    __shared__ int lds[4];
    const int i = ids[blockIdx.y];
    int n_good = 0;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < cap; j += gridDim.x * 256) {
        if (j == i) continue;
        const size_t k = rg_at(i, j, cap);
        if (st[k] == RG_NONE) continue;
        const float d = rg_dist(pos, i, j);
        float mx = maxd[k], mn = mind[k];
        if (d > mx) mx = d;
        if (d < mn) mn = d;
        maxd[k] = mx; mind[k] = mn;
        if (fabsf((mx - mn) / mn) > stretch_th) st[k] = (uint8_t)NRS_GRAPH_BAD;
        else ++n_good;
    }
    for (int off = 32; off > 0; off >>= 1) n_good += __shfl_xor(n_good, off, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = n_good;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&good[blockIdx.y], lds[0] + lds[1] + lds[2] + lds[3]);      // integer: order-free
}

// The same update when a large part of the points is listed (a frame's inliers: most of the map): one pass over the stored
// upper triangle, 64-row x 256-column tiles, a thread per column -- every pair that has a listed end is updated ONCE, by row-wise
// (coalesced) accesses; the kernel above reads the j < i part of a vertex's connections down a column (one 4-byte word per
// 4 * cap bytes) and updates a pair of two listed vertices twice.  slot[v] = place of v in the id list (-1: not listed, no
// duplicates); the counts are integer sums: order-free.  Same values: the distance is symmetric to the last bit.
constexpr int RG_TR = 64, RG_TC = 256;
__global__ __launch_bounds__(RG_TC) void k_rg_update_tri(const int* __restrict__ slot, const float* __restrict__ pos, int cap, float* maxd,
                                                         float* mind, uint8_t* st, float stretch_th, int* good) {
#pragma clang fp contract(off)
    __shared__ float rpos[RG_TR][3];
    __shared__ int rslot[RG_TR], rcnt[RG_TR];
    const int r0 = blockIdx.y * RG_TR, c0 = blockIdx.x * RG_TC, t = threadIdx.x;
    if (c0 + RG_TC - 1 <= r0) return;                               // (all of the tile is at or below the diagonal)
    if (t < RG_TR) {
        const int r = r0 + t;
        rslot[t] = r < cap ? slot[r] : -1;
        rcnt[t] = 0;
        for (int k = 0; k < 3; ++k) rpos[t][k] = r < cap ? pos[3 * r + k] : 0.f;
    }
    __syncthreads();
    const int j = c0 + t;
    const bool jin = j < cap;
    const int uj = jin ? slot[j] : -1;
    const float px = jin ? pos[3 * j] : 0.f, py = jin ? pos[3 * j + 1] : 0.f, pz = jin ? pos[3 * j + 2] : 0.f;
    int ccnt = 0;
    const int nr = min(RG_TR, cap - r0);
    for (int rr = 0; rr < nr; ++rr) {
        const int r = r0 + rr, ur = rslot[rr];
        bool pass = false;
        if (jin && j > r && (ur >= 0 || uj >= 0)) {
            const size_t k = (size_t)r * cap + j;
            if (st[k] != RG_NONE) {
                const float dx = rpos[rr][0] - px, dy = rpos[rr][1] - py, dz = rpos[rr][2] - pz;
                const float d = sqrtf(dx * dx + dy * dy + dz * dz);
                float mx = maxd[k], mn = mind[k];
                if (d > mx) mx = d;
                if (d < mn) mn = d;
                maxd[k] = mx; mind[k] = mn;
                if (fabsf((mx - mn) / mn) > stretch_th) st[k] = (uint8_t)NRS_GRAPH_BAD;
                else pass = true;
            }
        }
        if (pass && uj >= 0) ++ccnt;
        if (ur >= 0) {                                              // (block-uniform)
            const unsigned long long m = __ballot(pass);
            if ((t & 63) == 0 && m) atomicAdd(&rcnt[rr], __popcll(m));
        }
    }
    if (uj >= 0 && ccnt) atomicAdd(&good[uj], ccnt);
    __syncthreads();
    if (t < RG_TR && rslot[t] >= 0 && rcnt[t]) atomicAdd(&good[rslot[t]], rcnt[t]);
}

// GetEdges (regularization_graph.cc:71-87) for one listed vertex per wave.  The sorted list is
// [status 0 by weight desc][status 1 ...]...; the loop breaks at the FIRST entry below min_weight, i.e. inside the
// lowest status class s* that holds any such entry: the result is every entry of the classes below s* plus the
// entries of s* at or above min_weight -- all of them at or above min_weight, ordered by (status, -weight, index).
// (ties: ascending index, the build's documented choice where std::sort leaves the order open)
// a staged connection as one sortable word: status (8 bits) | inverted weight bits (32) | index (24): ascending keys =
// (status asc, weight desc, index asc)
__device__ inline unsigned long long rg_key(int s, float w, int j) {
    return ((unsigned long long)s << 56) | ((unsigned long long)(0xFFFFFFFFu - __float_as_uint(w)) << 24) | (unsigned long long)j;
}
// The state is the upper triangle of cap x cap arrays (rg_at): row i of the graph is a contiguous run for j > i and a COLUMN for
// j < i (one cache line per entry).  GetEdges of most rows (a2's first call: every tracked point) read that column part at a tenth of
// the memory system's rate (0.55 ms at 4.4k points), so the two arrays it scans are first written out as full symmetric matrices:
// 64 x 64 tiles, the transposed half through LDS, both halves stored as rows (~5 bytes x cap^2 read and written once).
__global__ __launch_bounds__(256) void k_rg_mirror(int cap, const uint8_t* __restrict__ st, const float* __restrict__ maxd, uint8_t* __restrict__ st_f, float* __restrict__ maxd_f) {
    __shared__ float tm[64][65];
    __shared__ uint8_t ts[64][68];
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bi > bj) return;                                           // (tiles of the upper triangle, the diagonal ones included)
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int i = bi * 64 + r, j = bj * 64 + tx;
        float v = 0.f; uint8_t q = RG_NONE;
        if (i < cap && j < cap && i < j) { v = maxd[(size_t)i * cap + j]; q = st[(size_t)i * cap + j]; }
        tm[r][tx] = v; ts[r][tx] = q;
        if (i < cap && j < cap && (bi < bj || i < j)) { maxd_f[(size_t)i * cap + j] = v; st_f[(size_t)i * cap + j] = q; }
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {                             // the transposed tile: rows of the lower triangle
        const int j = bj * 64 + r, i = bi * 64 + tx;               // (element (j, i) of the full matrix = element (i, j) of the state, i < j)
        if (i < cap && j < cap && i < j) { maxd_f[(size_t)j * cap + i] = tm[tx][r]; st_f[(size_t)j * cap + i] = ts[tx][r]; }
    }
}

constexpr int RG_BINS = 256;     // weight histogram bins per status class (selection of the first out_cap entries)
constexpr int RG_CLASSES = 4;    // NRS_GRAPH_VERIFIED .. NRS_GRAPH_BAD
constexpr int RG_SLACK = 1024;   // candidates staged beyond out_cap (the population of the bin the cut falls into)

// Only the first out_cap entries of the list are wanted (the callers walk a prefix), and with the reference's sigma most
// of the N - 1 connections survive the min_weight cut.  So the row is read twice: pass 1 finds s* and a 256-bin histogram
// of the surviving weights per status class, from which the (class, bin) at which the first out_cap entries end is
// known; pass 2 stages only the entries up to that bin (<= out_cap + one bin's population) and those are put in order.
// `select_all`: stage every survivor (the fallback when a bin holds more than RG_SLACK entries: many equal distances).
// `skip` (may be null): points the caller's walk passes over without any effect (the embedded mode's optimised points that
// carry no vertex, OPT:255-279 as restated in oracle/embedded_oracle.py): their connections are left out of the list unless
// BAD (a BAD connection ends the walk whoever it leads to).  They still count for s*: the list ends where the reference's ends.
__global__ __launch_bounds__(64) void k_rg_get_edges(int n_ids, const int* __restrict__ ids, int cap, const float* __restrict__ maxd,
                                                     const float* __restrict__ d0, const uint8_t* __restrict__ st, float sigma,
                                                     float min_w, float d_hi, int cand_cap, int out_cap, int select_all, const uint8_t* __restrict__ skip, int* o_count,
                                                     int* o_col, float* o_w, float* o_d0, int* o_st, int* overflow,
                                                     const uint8_t* __restrict__ st_f, const float* __restrict__ maxd_f) {   // st_f / maxd_f: the full matrices of k_rg_mirror, or null
    extern __shared__ unsigned long long cand[];              // keys of the staged connections (padded to a power of two for the sort)
    __shared__ int hist[RG_CLASSES][RG_BINS];
    const int r = blockIdx.x, lane = threadIdx.x;
    if (r >= n_ids) return;
    const int i = ids[r];
    for (int q = lane; q < RG_CLASSES * RG_BINS; q += 64) (&hist[0][0])[q] = 0;
    __syncthreads();
    const float bin_scale = (float)RG_BINS / (1.f - min_w);
    auto entry = [&](int j, float& w, int& s) -> bool {         // does connection (i, j) survive the min_weight cut?
        s = 255; w = 0.f;
        if (j >= cap || j == i) return false;
        const size_t k = st_f ? (size_t)i * cap + j : rg_at(i, j, cap);
        s = st_f ? st_f[k] : st[k];
        if (s == RG_NONE) { s = 255; return false; }
        const float mx = st_f ? maxd_f[k] : maxd[k];
        // beyond d_hi = 1.5 sigma (1 + 1e-4) the weight is below min_weight for sure; inside, the exact value decides
        if (mx <= d_hi) { w = rg_weight(mx, sigma); return !(w < min_w); }
        return false;
    };
    auto bin_of = [&](float w) { return min(RG_BINS - 1, max(0, (int)((1.f - w) * bin_scale))); };   // monotone in -w
    // ---- pass 1: s* (lowest status class holding an entry below min_weight) and the histograms
    int s_star = 255;
    for (int j0 = 0; j0 < cap; j0 += 64) {
        float w; int s;
        const bool keep = entry(j0 + lane, w, s);
        if (!keep && s != 255) s_star = min(s_star, s);
        if (keep && !(skip && s != NRS_GRAPH_BAD && skip[j0 + lane])) atomicAdd(&hist[min(s, RG_CLASSES - 1)][bin_of(w)], 1);
    }
    for (int off = 32; off > 0; off >>= 1) s_star = min(s_star, __shfl_xor(s_star, off, 64));
    __syncthreads();
    // ---- where do the first out_cap entries end?  cut[c] = last bin of class c that is staged (-1: none)
    int cut[RG_CLASSES];
    int n_out = 0, taken = 0;
#pragma unroll
    for (int c = 0; c < RG_CLASSES; ++c) {
        int loc[RG_BINS / 64], mine = 0;
#pragma unroll
        for (int q = 0; q < RG_BINS / 64; ++q) { loc[q] = hist[c][lane * (RG_BINS / 64) + q]; mine += loc[q]; }
        int incl = mine;                                         // inclusive scan over the lanes
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
        const int total = __shfl(incl, 63, 64);
        cut[c] = -1;
        if (c > s_star || total == 0) continue;
        n_out += total;
        const int rem = out_cap - taken;
        if (select_all || total <= rem) { cut[c] = RG_BINS - 1; taken += total; continue; }
        if (rem <= 0) continue;
        // first bin at which the running count reaches rem
        int run = incl - mine, hit = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < RG_BINS / 64; ++q) { run += loc[q]; if (run >= rem && hit == 0x7fffffff) hit = lane * (RG_BINS / 64) + q; }
        for (int off = 32; off > 0; off >>= 1) hit = min(hit, __shfl_xor(hit, off, 64));
        cut[c] = hit;
        taken = out_cap;
    }
    // ---- pass 2: stage the entries up to the cut (wave-ordered compaction; the counter is wave-uniform)
    int n = 0;
    for (int j0 = 0; j0 < cap; j0 += 64) {
        float w; int s;
        bool keep = entry(j0 + lane, w, s);
        if (keep && skip && s != NRS_GRAPH_BAD && skip[j0 + lane]) keep = false;
        if (keep) {
            const int c = min(s, RG_CLASSES - 1);
            int cc = cut[0];
#pragma unroll
            for (int q = 1; q < RG_CLASSES; ++q) cc = c == q ? cut[q] : cc;
            keep = s <= s_star && bin_of(w) <= cc;
        }
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const int p = n + __popcll(m & ((1ull << lane) - 1ull));
            if (p < cand_cap) cand[p] = rg_key(s, w, j0 + lane);
        }
        n += __popcll(m);
    }
    __syncthreads();
    if (n > cand_cap) { if (lane == 0) { atomicMax(overflow, n); o_count[r] = -1; } return; }
    // the survivors (status <= s*) in the order (status asc, weight desc, index asc); only the first out_cap are
    // written (callers walk a prefix: OPT:255-279 stops after 11 accepted neighbours or at the first BAD edge),
    // o_count is the full length (n_out, from the histograms)
    auto emit = [&](int rank, unsigned long long key) {
        const int j = (int)(key & 0xFFFFFFull);
        const size_t o = (size_t)r * out_cap + rank;
        o_col[o] = j; o_w[o] = __uint_as_float(0xFFFFFFFFu - (unsigned)((key >> 24) & 0xFFFFFFFFull)); o_st[o] = (int)(key >> 56);
        o_d0[o] = d0[rg_at(i, j, cap)];
    };
    if (n <= 384) {
        // short lists: rank sort, O(n^2 / 64)
        for (int a0 = 0; a0 < n; a0 += 64) {
            const int a = a0 + lane;
            if (a < n) {
                const unsigned long long ka = cand[a];
                int rank = 0;
                for (int b = 0; b < n; ++b) rank += cand[b] < ka ? 1 : 0;
                if (rank < out_cap) emit(rank, ka);
            }
        }
    } else {
        // long lists (a sigma that covers most of the map): bitonic sort of the staged keys in LDS, one wave
        int N = 512;
        while (N < n) N <<= 1;
        for (int a = n + lane; a < N; a += 64) cand[a] = ~0ull;
        __syncthreads();
        for (int k = 2; k <= N; k <<= 1)
            for (int jj = k >> 1; jj > 0; jj >>= 1) {
                for (int t = lane; t < N / 2; t += 64) {
                    const int lo = ((t & ~(jj - 1)) << 1) | (t & (jj - 1)), hi = lo | jj;    // the t-th pair at distance jj
                    const unsigned long long x = cand[lo], y = cand[hi];
                    const bool up = (lo & k) == 0;
                    if ((x > y) == up) { cand[lo] = y; cand[hi] = x; }
                }
                __syncthreads();
            }
        for (int a = lane; a < min(n, out_cap); a += 64) emit(a, cand[a]);
    }
    if (lane == 0) o_count[r] = n_out;
}

__global__ void k_rg_fill(uint8_t* st, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) st[i] = RG_NONE;
}

static int rg_check_ids(nrs_rgraph* g, int n, const int32_t* ids, const char* what) {
    if (n < 0 || (n > 0 && !ids)) return g->c->fail(NRS_ERR_INVALID, "%s: null / negative id list", what);
    for (int i = 0; i < n; ++i)
        if (ids[i] < 0 || ids[i] >= g->cap) return g->c->fail(NRS_ERR_INVALID, "%s: point index %d outside [0, %d)", what, ids[i], g->cap);
    return NRS_OK;
}

}  // namespace nrs

using namespace nrs;

extern "C" int nrs_rgraph_create(nrs_ctx* c, int32_t capacity, float sigma, float stretch_th, nrs_rgraph** out) {
    if (!c || !out) return NRS_ERR_INVALID;
    *out = nullptr;
    if (capacity <= 1 || capacity > 200000 || !(sigma > 0) || !(stretch_th > 0)) return c->fail(NRS_ERR_INVALID, "nrs_rgraph_create: bad argument");
    NRS_HIP(c, hipSetDevice(c->device));
    nrs_rgraph* g = new (std::nothrow) nrs_rgraph();
    if (!g) return c->fail(NRS_ERR_ALLOC, "out of host memory");
    g->c = c; g->cap = capacity; g->stretch_th = stretch_th;
    g->sigma = sigma;
    g->min_w = rg_weight((float)((double)sigma * 1.5), sigma);          // regularization_graph.cc:29 (float * double literal -> float argument)
    const size_t n2 = (size_t)capacity * capacity;
    hipError_t e = hipMalloc((void**)&g->maxd, n2 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&g->mind, n2 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&g->d0, n2 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&g->st, n2);
    if (e != hipSuccess) {
        const int rc = c->fail(NRS_ERR_ALLOC, "nrs_rgraph_create: %zu bytes for %d points: %s", n2 * 13, capacity, hipGetErrorString(e));
        if (g->maxd) (void)hipFree(g->maxd);
        if (g->mind) (void)hipFree(g->mind);
        if (g->d0) (void)hipFree(g->d0);
        delete g;
        return rc;
    }
    hipLaunchKernelGGL(k_rg_fill, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, c->stream, g->st, n2);
    NRS_HIP(c, hipGetLastError());
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    *out = g;
    return NRS_OK;
}

extern "C" void nrs_rgraph_destroy(nrs_rgraph* g) {
    if (!g) return;
    nrs_ctx* c = g->c;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(g->maxd); (void)hipFree(g->mind); (void)hipFree(g->d0); (void)hipFree(g->st);
    if (g->pin) (void)hipHostFree(g->pin);
    c->release(g->pos); c->release(g->ids_a); c->release(g->ids_b); c->release(g->out_i); c->release(g->out_f); c->release(g->good); c->release(g->skip); c->release(g->slot); c->release(g->walk); c->release(g->mir);
    delete g;
}

extern "C" int nrs_rgraph_set_sigma(nrs_rgraph* g, float sigma) {
    if (!g) return NRS_ERR_INVALID;
    if (!(sigma > 0)) return g->c->fail(NRS_ERR_INVALID, "sigma must be positive");
    g->sigma = sigma;
    g->min_w = rg_weight((float)((double)sigma * 1.5), sigma);          // RegularizationGraph::SetSigma (:33-36)
    return NRS_OK;
}

extern "C" float nrs_rgraph_min_weight(const nrs_rgraph* g) { return g ? g->min_w : 0.f; }

extern "C" int nrs_rgraph_add_edges(nrs_rgraph* g, const float* pos, int32_t n_new, const int32_t* new_ids, int32_t n_other,
                                    const int32_t* other_ids) {
    if (!g) return NRS_ERR_INVALID;
    nrs_ctx* c = g->c;
    if (!pos) return c->fail(NRS_ERR_INVALID, "nrs_rgraph_add_edges: null positions");
    NRS_TRY(rg_check_ids(g, n_new, new_ids, "nrs_rgraph_add_edges"));
    NRS_TRY(rg_check_ids(g, n_other, other_ids, "nrs_rgraph_add_edges"));
    if (n_new == 0 || n_other == 0) return NRS_OK;
    NRS_HIP(c, hipSetDevice(c->device));
    NRS_TRY(c->ensure(g->pos, sizeof(float) * 3 * (size_t)g->cap));
    NRS_TRY(c->ensure(g->ids_a, sizeof(int) * (size_t)n_new));
    NRS_TRY(c->ensure(g->ids_b, sizeof(int) * (size_t)n_other));
    NRS_HIP(c, hipMemcpyAsync(g->pos.p, pos, sizeof(float) * 3 * (size_t)g->cap, hipMemcpyHostToDevice, c->stream));
    NRS_HIP(c, hipMemcpyAsync(g->ids_a.p, new_ids, sizeof(int) * (size_t)n_new, hipMemcpyHostToDevice, c->stream));
    NRS_HIP(c, hipMemcpyAsync(g->ids_b.p, other_ids, sizeof(int) * (size_t)n_other, hipMemcpyHostToDevice, c->stream));
    for (int a0 = 0; a0 < n_new; a0 += 32768) {                   // grid.y limit
        const int na = std::min(32768, n_new - a0);
        hipLaunchKernelGGL(k_rg_add, dim3((n_other + 255) / 256, na), dim3(256), 0, c->stream, na, g->ids_a.as<int>() + a0, n_other,
                           g->ids_b.as<int>(), g->pos.as<float>(), g->cap, g->maxd, g->mind, g->d0, g->st);
    }
    NRS_HIP(c, hipGetLastError());
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    return NRS_OK;
}

extern "C" int nrs_rgraph_update(nrs_rgraph* g, const float* pos, int32_t n_ids, const int32_t* ids, int32_t* good_count) {
    if (!g) return NRS_ERR_INVALID;
    nrs_ctx* c = g->c;
    if (!pos || (n_ids > 0 && !good_count)) return c->fail(NRS_ERR_INVALID, "nrs_rgraph_update: null argument");
    NRS_TRY(rg_check_ids(g, n_ids, ids, "nrs_rgraph_update"));
    if (n_ids == 0) return NRS_OK;
    NRS_HIP(c, hipSetDevice(c->device));
    NRS_TRY(c->ensure(g->pos, sizeof(float) * 3 * (size_t)g->cap));
    NRS_TRY(c->ensure(g->ids_a, sizeof(int) * (size_t)n_ids));
    NRS_TRY(c->ensure(g->good, sizeof(int) * (size_t)n_ids));
    NRS_HIP(c, hipMemcpyAsync(g->pos.p, pos, sizeof(float) * 3 * (size_t)g->cap, hipMemcpyHostToDevice, c->stream));
    NRS_HIP(c, hipMemcpyAsync(g->ids_a.p, ids, sizeof(int) * (size_t)n_ids, hipMemcpyHostToDevice, c->stream));
    NRS_HIP(c, hipMemsetAsync(g->good.p, 0, sizeof(int) * (size_t)n_ids, c->stream));
    // most of the points listed, each once: the one-pass form
    bool tri = (size_t)n_ids * 4 >= (size_t)g->cap;
    if (tri) {
        g->h_slot.assign(g->cap, -1);
        for (int a = 0; a < n_ids && tri; ++a) { if (g->h_slot[ids[a]] >= 0) tri = false; g->h_slot[ids[a]] = a; }
    }
    if (tri) {
        NRS_TRY(c->ensure(g->slot, sizeof(int) * (size_t)g->cap));
        NRS_HIP(c, hipMemcpyAsync(g->slot.p, g->h_slot.data(), sizeof(int) * (size_t)g->cap, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_rg_update_tri, dim3((g->cap + RG_TC - 1) / RG_TC, (g->cap + RG_TR - 1) / RG_TR), dim3(RG_TC), 0, c->stream,
                           g->slot.as<int>(), g->pos.as<float>(), g->cap, g->maxd, g->mind, g->st, g->stretch_th, g->good.as<int>());
    }
    const int bx = std::max(1, std::min(8, (g->cap + 1023) / 1024));
    for (int a0 = 0; a0 < n_ids && !tri; a0 += 32768) {
        const int na = std::min(32768, n_ids - a0);
        hipLaunchKernelGGL(k_rg_update, dim3(bx, na), dim3(256), 0, c->stream, na, g->ids_a.as<int>() + a0, g->pos.as<float>(), g->cap,
                           g->maxd, g->mind, g->st, g->stretch_th, g->good.as<int>() + a0);
    }
    NRS_HIP(c, hipGetLastError());
    NRS_HIP(c, hipMemcpyAsync(good_count, g->good.p, sizeof(int) * (size_t)n_ids, hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    return NRS_OK;
}

// GetEdges of the listed points into the graph's pinned staging area: count[n_ids] | col | status | weight | first distance
// (each n_ids x cap_per_point).  nrs_rgraph_get_edges copies from there; the a2 driver reads it in place.
namespace nrs {
int rg_capacity(const nrs_rgraph* g) { return g->cap; }
// the longest GetEdges prefix a call can ask for: the candidates of a row are sorted in LDS
int rg_max_cap_per_point(const nrs_rgraph* g) {
    int lds_max = 65536;
    (void)hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, g->c->device);
    size_t cand_max = 512;
    while (sizeof(unsigned long long) * (cand_max << 1) + 4096 + 512 <= (size_t)lds_max) cand_max <<= 1;
    return std::min(g->cap, (int)cand_max - RG_SLACK);
}

// ---- a2's neighbour walk on the device (OPT:252-279 as nrs_track.hip's host loop restates it): optimised point idx (rows of the last
// GetEdges, in the caller's order) accepts, in list order, up to 11 connections whose other end takes part (code >= 0: its index among
// the optimised points) and -- for a vertex -- is not already paired with it, i.e. NOT (io < idx and idx among io's accepted): a
// duplicate does not count, so a point's accepted set depends on those of lower-index points.  The sequential loop is the unique
// solution of that recursion; the kernel recomputes every point's set from the current sets of the others (one wave per point, 64 list
// entries at a time) and is repeated until a pass changes nothing (a pass without a change is a fixed point; correct sets never change
// again, so the passes needed are the longest chain of dependent points + 1).  The last pass (final = 1) also leaves weights, first
// distances, the lost-point flags of the entries the walk visits and whether it ended before its list did.
constexpr int RG_WALK_MAX = 11;
__global__ __launch_bounds__(256) void k_rg_walk(int n, int cap, const int* __restrict__ cnt, const int* __restrict__ col, const int* __restrict__ st,
                                                 const float* __restrict__ w, const float* __restrict__ d0, const int* __restrict__ code,
                                                 const uint8_t* __restrict__ is_node, int* acc, int* n_acc, float* acc_w, float* acc_d0,
                                                 uint8_t* ended, uint8_t* lost, int* changed, int final) {
    __shared__ int s_new[4][RG_WALK_MAX], s_pos[4][RG_WALK_MAX];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int idx = blockIdx.x * 4 + wv;
    if (idx >= n) return;
    const int c = min(cnt[idx], cap);
    const size_t base = (size_t)idx * cap;
    const bool node = !is_node || is_node[idx];
    int n_reg = 0, brk = -1;                                       // brk: the entry at which the sequential loop breaks (-1: it runs off the list)
    for (int ch = 0; ch < c && brk < 0; ch += 64) {
        const int a = ch + lane;
        const bool valid = a < c;
        const int other = valid ? col[base + a] : 0;
        const bool bad = valid && st[base + a] == NRS_GRAPH_BAD;
        const int io = valid ? code[other] : -1;
        bool keep = io >= 0;
        if (keep && node && io < idx) {
            const int na = min(n_acc[io], RG_WALK_MAX);
            for (int k = 0; k < na; ++k) keep = keep && acc[(size_t)io * RG_WALK_MAX + k] != idx;
        }
        const unsigned long long badm = __ballot(bad), keepm = __ballot(keep);
        const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
        const int rank = __popcll(keepm & below);                  // kept entries of this chunk before this lane
        // the loop breaks at the first entry that is BAD or finds eleven accepted before it
        const bool brk_here = valid && (bad || n_reg + rank >= RG_WALK_MAX);
        const unsigned long long brkm = __ballot(brk_here);
        const int first_brk = brkm ? __ffsll((long long)brkm) - 1 : 64;
        if (keep && lane < first_brk) { s_new[wv][n_reg + rank] = io; s_pos[wv][n_reg + rank] = a; }
        if (final && valid && lane < first_brk && io == -2) lost[other] = 1;
        n_reg += __popcll(keepm & (first_brk < 64 ? (~0ull >> (63 - first_brk) >> 1) : ~0ull));
        if (first_brk < 64) brk = ch + first_brk;
        (void)badm;
    }
    __builtin_amdgcn_wave_barrier();
    bool diff = false;
    const int old_n = n_acc[idx];
    if (lane < n_reg) diff = acc[(size_t)idx * RG_WALK_MAX + lane] != s_new[wv][lane];
    diff = __any(diff) || old_n != n_reg;
    if (diff) {
        if (lane < n_reg) acc[(size_t)idx * RG_WALK_MAX + lane] = s_new[wv][lane];
        __threadfence();
        if (lane == 0) { n_acc[idx] = n_reg; atomicAdd(changed, 1); }
    }
    if (final) {
        if (lane < n_reg) { acc_w[(size_t)idx * RG_WALK_MAX + lane] = w[base + s_pos[wv][lane]]; acc_d0[(size_t)idx * RG_WALK_MAX + lane] = d0[base + s_pos[wv][lane]]; }
        if (lane == 0) ended[idx] = brk >= 0;
    }
}

// The walk over the lists of the last rg_get_edges_staged(.., lists_to_host = false) call.  code: n_map ints (nrs_track.hip walk_code);
// is_node: one byte per row or null (every optimised point a vertex).  Outputs (host): n_acc[n], acc[11 n] (indices among the optimised
// points), acc_w / acc_d0[11 n], ended[n], lost[n_map].  *converged = 0: the passes ran out (the caller walks on the host instead).
int rg_walk(nrs_rgraph* g, int n_map, const int* code, const uint8_t* is_node, int* n_acc, int* acc, float* acc_w, float* acc_d0,
            uint8_t* ended, uint8_t* lost, int* converged, int* passes) {
    nrs_ctx* c = g->c;
    const int n = g->last_n_ids, cap = g->last_cap;
    *converged = 0; *passes = 0;
    if (n <= 0 || n_map > g->cap) return c->fail(NRS_ERR_STATE, "rg_walk: no lists on the device");
    const size_t no = (size_t)n * cap;
    const int* d_cnt = g->out_i.as<int>();
    const int* d_col = d_cnt + n;
    const int* d_st = d_col + no;
    const float* d_w = reinterpret_cast<const float*>(d_st + no);
    const float* d_d0 = d_w + no;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    constexpr int MAXP = 96;
    const size_t o_code = 0, o_node = o_code + al(4 * (size_t)n_map), o_acc = o_node + al((size_t)n), o_nacc = o_acc + al(4 * (size_t)RG_WALK_MAX * n),
                 o_w = o_nacc + al(4 * (size_t)n), o_d0 = o_w + al(4 * (size_t)RG_WALK_MAX * n), o_end = o_d0 + al(4 * (size_t)RG_WALK_MAX * n),
                 o_lost = o_end + al((size_t)n), o_chg = o_lost + al((size_t)n_map), total = o_chg + al(4 * (MAXP + 1));
    NRS_TRY(c->ensure(g->walk, total));
    char* b = g->walk.as<char>();
    NRS_HIP(c, hipMemcpyAsync(b + o_code, code, 4 * (size_t)n_map, hipMemcpyHostToDevice, c->stream));
    if (is_node) NRS_HIP(c, hipMemcpyAsync(b + o_node, is_node, (size_t)n, hipMemcpyHostToDevice, c->stream));
    NRS_HIP(c, hipMemsetAsync(b + o_nacc, 0, 4 * (size_t)n, c->stream));
    NRS_HIP(c, hipMemsetAsync(b + o_lost, 0, o_chg - o_lost + 4 * (MAXP + 1), c->stream));
    int* chg = reinterpret_cast<int*>(b + o_chg);
    auto pass = [&](int slot, int final) {
        hipLaunchKernelGGL(k_rg_walk, dim3((n + 3) / 4), dim3(256), 0, c->stream, n, cap, d_cnt, d_col, d_st, d_w, d_d0, reinterpret_cast<const int*>(b + o_code),
                           is_node ? reinterpret_cast<const uint8_t*>(b + o_node) : nullptr, reinterpret_cast<int*>(b + o_acc), reinterpret_cast<int*>(b + o_nacc),
                           reinterpret_cast<float*>(b + o_w), reinterpret_cast<float*>(b + o_d0), reinterpret_cast<uint8_t*>(b + o_end),
                           reinterpret_cast<uint8_t*>(b + o_lost), chg + slot, final);
    };
    int done = 0, hc[MAXP];
    int max_p = MAXP;                                              // (NRS_WALK_MAX_PASSES: a lower cap, so that the tests reach the caller's host fallback)
    if (const char* v = c->env("NRS_WALK_MAX_PASSES")) max_p = std::max(1, std::min(MAXP, atoi(v)));
    while (done < max_p && !*converged) {                          // batches of passes, one look at their change counts per batch
        const int batch = std::min(done == 0 ? 6 : 8, max_p - done);
        for (int q = 0; q < batch; ++q) pass(done + q, 0);
        NRS_HIP(c, hipGetLastError());
        NRS_HIP(c, hipMemcpyAsync(hc + done, chg + done, 4 * (size_t)batch, hipMemcpyDeviceToHost, c->stream));
        NRS_HIP(c, hipStreamSynchronize(c->stream));
        for (int q = 0; q < batch; ++q) if (hc[done + q] == 0) *converged = 1;
        done += batch;
    }
    *passes = done;
    if (!*converged) return NRS_OK;
    pass(MAXP, 1);                                                 // (the sets are final: this pass changes nothing and leaves the outputs)
    NRS_HIP(c, hipGetLastError());
    NRS_HIP(c, hipMemcpyAsync(n_acc, b + o_nacc, 4 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipMemcpyAsync(acc, b + o_acc, 4 * (size_t)RG_WALK_MAX * n, hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipMemcpyAsync(acc_w, b + o_w, 4 * (size_t)RG_WALK_MAX * n, hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipMemcpyAsync(acc_d0, b + o_d0, 4 * (size_t)RG_WALK_MAX * n, hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipMemcpyAsync(ended, b + o_end, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipMemcpyAsync(lost, b + o_lost, (size_t)n_map, hipMemcpyDeviceToHost, c->stream));
    int last = 0;
    NRS_HIP(c, hipMemcpyAsync(&last, chg + MAXP, 4, hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    if (last != 0) *converged = 0;                                 // (cannot happen: the fixed point was reached)
    return NRS_OK;
}

int rg_get_edges_staged(nrs_rgraph* g, int32_t n_ids, const int32_t* ids, int32_t cap_per_point, const int** count, const int** col,
                        const int** status, const float** w, const float** d0, const uint8_t* pass_over, bool lists_to_host) {   // lists_to_host = false: the lists stay on the device (rg_walk reads them there); only the counts come back
    nrs_ctx* c = g->c;
    NRS_TRY(rg_check_ids(g, n_ids, ids, "GetEdges"));               // (also the a2 driver's entry: ids index the dense state)
    if (cap_per_point <= 0) return c->fail(NRS_ERR_INVALID, "GetEdges: cap_per_point must be positive");
    NRS_HIP(c, hipSetDevice(c->device));
    // the sort buffer of a row lives in LDS: 8 bytes per staged candidate (a power of two of them) + 4 KB of histogram
    int lds_max = 0;
    NRS_HIP(c, hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, c->device));
    size_t cand_max = 512;
    while (sizeof(unsigned long long) * (cand_max << 1) + 4096 + 512 <= (size_t)lds_max) cand_max <<= 1;
    if ((size_t)std::min(g->cap, cap_per_point + RG_SLACK) > cand_max)
        return c->fail(NRS_ERR_INVALID, "GetEdges: cap_per_point %d needs more than the %zu candidates per row that fit this device's %d bytes of LDS",
                       cap_per_point, cand_max, lds_max);
    const size_t no = (size_t)n_ids * cap_per_point;
    NRS_TRY(c->ensure(g->ids_a, sizeof(int) * (size_t)n_ids));
    NRS_TRY(c->ensure(g->out_i, sizeof(int) * (4 * no + (size_t)n_ids + 4)));
    // one device block in the staging layout: count | col | status | weight | first distance | overflow word
    int* d_cnt = g->out_i.as<int>();
    int* d_col = d_cnt + n_ids;
    int* d_st = d_col + no;
    float* d_w = reinterpret_cast<float*>(d_st + no);
    float* d_d0 = d_w + no;
    int* d_ovf = reinterpret_cast<int*>(d_d0 + no);
    const size_t bytes = sizeof(int) * (4 * no + (size_t)n_ids + 1);
    if (bytes > g->pin_cap) {
        if (g->pin) (void)hipHostFree(g->pin);
        g->pin = nullptr; g->pin_cap = 0;
        const size_t want = bytes + bytes / 4;
        if (hipHostMalloc((void**)&g->pin, want, hipHostMallocDefault) != hipSuccess) return c->fail(NRS_ERR_ALLOC, "nrs_rgraph_get_edges: %zu bytes of pinned host memory", want);
        g->pin_cap = want;
    }
    NRS_HIP(c, hipMemcpyAsync(g->ids_a.p, ids, sizeof(int) * (size_t)n_ids, hipMemcpyHostToDevice, c->stream));
    if (pass_over) {                                               // (one byte per point of the dense state)
        NRS_TRY(c->ensure(g->skip, (size_t)g->cap));
        NRS_HIP(c, hipMemcpyAsync(g->skip.p, pass_over, (size_t)g->cap, hipMemcpyHostToDevice, c->stream));
    }
    const float d_hi = (float)((double)g->sigma * 1.5 * (1.0 + 1e-4));
    // many rows (a2's first GetEdges of a frame: every tracked point): the kernel scans full symmetric copies of the two arrays it reads
    const uint8_t* st_f = nullptr;
    const float* maxd_f = nullptr;
    if ((int64_t)n_ids * 8 >= g->cap && g->cap >= 512 && !c->env("NRS_RG_NO_MIRROR")) {
        const size_t cc = (size_t)g->cap * g->cap, o_m = (cc + 255) & ~(size_t)255;
        if (c->ensure(g->mir, o_m + 4 * cc) == NRS_OK) {           // (5 bytes per pair on top of the 13 of the state)
            const int nt = (g->cap + 63) / 64;
            hipLaunchKernelGGL(k_rg_mirror, dim3(nt, nt), dim3(256), 0, c->stream, g->cap, g->st, g->maxd, g->mir.as<uint8_t>(), reinterpret_cast<float*>(g->mir.as<char>() + o_m));
            NRS_HIP(c, hipGetLastError());
            st_f = g->mir.as<uint8_t>(); maxd_f = reinterpret_cast<const float*>(g->mir.as<char>() + o_m);
        } else {                                                   // no memory for the copies: the triangular state serves the call (slower, same lists)
            (void)hipGetLastError();
            c->err[0] = 0;
        }
    }
    // first the selecting form (stages <= cap_per_point + one histogram bin per row); if a bin overflows the staging area
    // (many equal distances), once more with every survivor of a row staged
    int* h = reinterpret_cast<int*>(g->pin);
    int cand_cap = 0;
    for (int pass = 0; pass < 2; ++pass) {
        cand_cap = pass == 0 ? std::min(g->cap, cap_per_point + RG_SLACK) : (int)std::min<size_t>({(size_t)g->cap, (size_t)12000, cand_max});
        NRS_HIP(c, hipMemsetAsync(d_ovf, 0, sizeof(int), c->stream));
        size_t pad = 512;                                          // the long-list path sorts a power-of-two number of keys
        while (pad < (size_t)cand_cap) pad <<= 1;
        const size_t shm = sizeof(unsigned long long) * pad;
        NRS_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_rg_get_edges), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        hipLaunchKernelGGL(k_rg_get_edges, dim3(n_ids), dim3(64), shm, c->stream, n_ids, g->ids_a.as<int>(), g->cap,
                           g->maxd, g->d0, g->st, g->sigma, g->min_w, d_hi, cand_cap, cap_per_point, pass, pass_over ? g->skip.as<uint8_t>() : nullptr, d_cnt, d_col, d_w, d_d0, d_st, d_ovf, st_f, maxd_f);
        NRS_HIP(c, hipGetLastError());
        if (lists_to_host) NRS_HIP(c, hipMemcpyAsync(h, d_cnt, bytes, hipMemcpyDeviceToHost, c->stream));
        else {
            NRS_HIP(c, hipMemcpyAsync(h, d_cnt, sizeof(int) * (size_t)n_ids, hipMemcpyDeviceToHost, c->stream));
            NRS_HIP(c, hipMemcpyAsync(h + 4 * no + (size_t)n_ids, d_ovf, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        }
        NRS_HIP(c, hipStreamSynchronize(c->stream));
        g->last_n_ids = n_ids; g->last_cap = cap_per_point;
        if (h[4 * no + (size_t)n_ids] == 0) break;
        if (pass == 1) return c->fail(NRS_ERR_INVALID, "nrs_rgraph_get_edges: a point has %d connections at or above min_weight: more than this build stages per row (%d)", h[4 * no + (size_t)n_ids], cand_cap);
    }
    *count = h;
    *col = h + n_ids;
    *status = h + n_ids + no;
    *w = reinterpret_cast<const float*>(h + n_ids + 2 * no);
    *d0 = reinterpret_cast<const float*>(h + n_ids + 3 * no);
    return NRS_OK;
}
}  // namespace nrs

extern "C" int nrs_rgraph_get_edges(nrs_rgraph* g, int32_t n_ids, const int32_t* ids, int32_t cap_per_point, int32_t* count,
                                    int32_t* col, float* w, float* d0, int32_t* status) {
    if (!g) return NRS_ERR_INVALID;
    nrs_ctx* c = g->c;
    if (cap_per_point <= 0 || (n_ids > 0 && (!count || !col || !w || !d0 || !status))) return c->fail(NRS_ERR_INVALID, "nrs_rgraph_get_edges: bad argument");
    NRS_TRY(rg_check_ids(g, n_ids, ids, "nrs_rgraph_get_edges"));
    if (n_ids == 0) return NRS_OK;
    const int *h_cnt, *h_col, *h_st;
    const float *h_w, *h_d0;
    NRS_TRY(rg_get_edges_staged(g, n_ids, ids, cap_per_point, &h_cnt, &h_col, &h_st, &h_w, &h_d0, nullptr, true));
    const size_t no = (size_t)n_ids * cap_per_point;
    memcpy(count, h_cnt, sizeof(int) * (size_t)n_ids);
    memcpy(col, h_col, sizeof(int) * no);
    memcpy(status, h_st, sizeof(int) * no);
    memcpy(w, h_w, sizeof(float) * no);
    memcpy(d0, h_d0, sizeof(float) * no);
    return NRS_OK;
}

// GetEdge (regularization_graph.cc:57-59) / parity tap: out = {weight, first_distance, max_distance, min_distance}; status -1 = no edge
extern "C" int nrs_rgraph_edge(nrs_rgraph* g, int32_t i, int32_t j, float out[4], int32_t* status) {
    if (!g) return NRS_ERR_INVALID;
    nrs_ctx* c = g->c;
    if (i < 0 || j < 0 || i >= g->cap || j >= g->cap || i == j || !out || !status) return c->fail(NRS_ERR_INVALID, "nrs_rgraph_edge: bad argument");
    NRS_HIP(c, hipSetDevice(c->device));
    const size_t k = i < j ? (size_t)i * g->cap + j : (size_t)j * g->cap + i;
    uint8_t s = 0;
    NRS_HIP(c, hipMemcpy(&s, g->st + k, 1, hipMemcpyDeviceToHost));
    NRS_HIP(c, hipMemcpy(&out[2], g->maxd + k, 4, hipMemcpyDeviceToHost));
    NRS_HIP(c, hipMemcpy(&out[3], g->mind + k, 4, hipMemcpyDeviceToHost));
    NRS_HIP(c, hipMemcpy(&out[1], g->d0 + k, 4, hipMemcpyDeviceToHost));
    out[0] = rg_weight(out[2], g->sigma);
    *status = s == RG_NONE ? -1 : (int32_t)s;
    return NRS_OK;
}

// parity tap: rows of the dense state (n_ids x capacity each; status 255 = no edge)
extern "C" int nrs_rgraph_rows(nrs_rgraph* g, int32_t n_ids, const int32_t* ids, float* maxd, float* mind, float* d0, uint8_t* status) {
    if (!g) return NRS_ERR_INVALID;
    nrs_ctx* c = g->c;
    NRS_TRY(rg_check_ids(g, n_ids, ids, "nrs_rgraph_rows"));
    NRS_HIP(c, hipSetDevice(c->device));
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    const int cap = g->cap;
    std::vector<float> col(cap);
    std::vector<uint8_t> colb(cap);
    for (int r = 0; r < n_ids; ++r) {
        const int i = ids[r];
        // row part (j > i) is contiguous; the column part (j < i) is strided in the canonical storage
        auto fetch = [&](const float* src, float* dst) -> int {
            if (!dst) return NRS_OK;
            NRS_HIP(c, hipMemcpy(dst + (size_t)r * cap + i, src + (size_t)i * cap + i, sizeof(float) * (size_t)(cap - i), hipMemcpyDeviceToHost));
            if (i > 0) NRS_HIP(c, hipMemcpy2D(dst + (size_t)r * cap, sizeof(float), src + i, sizeof(float) * (size_t)cap, sizeof(float), (size_t)i, hipMemcpyDeviceToHost));
            return NRS_OK;
        };
        NRS_TRY(fetch(g->maxd, maxd)); NRS_TRY(fetch(g->mind, mind)); NRS_TRY(fetch(g->d0, d0));
        if (status) {
            NRS_HIP(c, hipMemcpy(status + (size_t)r * cap + i, g->st + (size_t)i * cap + i, (size_t)(cap - i), hipMemcpyDeviceToHost));
            if (i > 0) NRS_HIP(c, hipMemcpy2D(status + (size_t)r * cap, 1, g->st + i, (size_t)cap, 1, (size_t)i, hipMemcpyDeviceToHost));
            status[(size_t)r * cap + i] = RG_NONE;
        }
    }
    return NRS_OK;
}
