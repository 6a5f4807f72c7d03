// N2, "5k points x 500 graph nodes": node selection for the skinned mode.
//
// The reference's deformation graph has no separate node set (every map point is a vertex).  Its own form of skinning is
// stage 2 of the pose-and-deformation optimisation (modules/optimization/g2o_optimization.cc:476-553): points of the frame
// that are NOT in the optimisation follow <= 11 optimised graph neighbours through SpatialRegularizerFixed
// (spatial_regularizer_fixed.cc:32-43).  The skinned mode is that algorithm with a chosen node set: M points (farthest
// point sampling) stay TRACKED_WITH_3D and carry the free variables of stage 1, the other points of the
// frame enter as the "lost" set of stage 2.  Nothing in the solve is new (nrs_track_deform_solve[_rg] runs it, the
// oracle's track_deform_solve checks it); what this file adds is the selection.
//
// Farthest point sampling, one workgroup: pick 0 = the eligible point with the lowest index; pick k = the eligible point
// farthest from the picks so far (fp32 squared distance (dx*dx + dy*dy) + dz*dz without contraction, ties: lowest index).
// Per round: every thread folds the last pick into the running minimum distance of its points (n / 1024 each, coalesced),
// then one arg-max over the workgroup (wave shuffles + LDS).  5k points x 500 nodes: 500 rounds of ~2 us.
#include <cmath>
#include <vector>
#include "nrs_ctx.hpp"

namespace nrs {

constexpr int FPS_BLK = 1024;

__global__ __launch_bounds__(FPS_BLK) void k_fps(int n, const float* __restrict__ pos, const uint8_t* __restrict__ eligible, int m,
                                                 float* mind, int* out_ids, int* n_out) {
#pragma clang fp contract(off)
    __shared__ float s_v[FPS_BLK / 64];
    __shared__ int s_i[FPS_BLK / 64];
    __shared__ int s_pick;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // round 0: lowest eligible index
    int first = 0x7fffffff;
    for (int j = tid; j < n; j += FPS_BLK) {
        const bool ok = !eligible || eligible[j];
        mind[j] = ok ? INFINITY : -1.f;
        if (ok) first = min(first, j);
    }
    for (int off = 32; off > 0; off >>= 1) first = min(first, __shfl_xor(first, off, 64));
    if (lane == 0) s_i[wave] = first;
    __syncthreads();
    if (tid == 0) {
        int f = s_i[0];
        for (int w = 1; w < FPS_BLK / 64; ++w) f = min(f, s_i[w]);
        s_pick = f;
    }
    __syncthreads();
    int pick = s_pick;
    int k = 0;
    while (pick != 0x7fffffff && k < m) {
        if (tid == 0) out_ids[k] = pick;
        ++k;
        if (k == m) break;
        const float px = pos[3 * (size_t)pick], py = pos[3 * (size_t)pick + 1], pz = pos[3 * (size_t)pick + 2];
        float bv = -1.f;
        int bi = 0x7fffffff;
        for (int j = tid; j < n; j += FPS_BLK) {
            float d = mind[j];
            if (j == pick) d = -1.f;
            else if (d >= 0.f) {
                const float dx = pos[3 * (size_t)j] - px, dy = pos[3 * (size_t)j + 1] - py, dz = pos[3 * (size_t)j + 2] - pz;
                const float d2 = (dx * dx + dy * dy) + dz * dz;
                d = d2 < d ? d2 : d;
            }
            mind[j] = d;
            if (d > bv) { bv = d; bi = j; }                        // ascending j per thread: the first maximum is kept
        }
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        __syncthreads();                                           // (s_v / s_i of the previous round have been read)
        if (lane == 0) { s_v[wave] = bv; s_i[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float v = s_v[0];
            int i = s_i[0];
            for (int w = 1; w < FPS_BLK / 64; ++w)
                if (s_v[w] > v || (s_v[w] == v && s_i[w] < i)) { v = s_v[w]; i = s_i[w]; }
            s_pick = v >= 0.f ? i : 0x7fffffff;                    // nothing eligible is left
        }
        __syncthreads();
        pick = s_pick;
    }
    if (tid == 0) *n_out = k;
}

}  // namespace nrs

using namespace nrs;

extern "C" int nrs_skin_select_nodes(nrs_ctx* c, int32_t n_points, const float* pos, const uint8_t* eligible, int32_t n_nodes,
                                     int32_t* node_ids) {
    if (!c) return NRS_ERR_INVALID;
    if (!pos || !node_ids || n_points <= 0 || n_nodes <= 0 || n_nodes > n_points) return c->fail(NRS_ERR_INVALID, "nrs_skin_select_nodes: bad argument");
    NRS_HIP(c, hipSetDevice(c->device));
    DevBuf d_pos, d_el, d_mind, d_out;
    struct Rel { nrs_ctx* c; DevBuf *a, *b, *d, *e; ~Rel() { c->release(*a); c->release(*b); c->release(*d); c->release(*e); } } rel{c, &d_pos, &d_el, &d_mind, &d_out};
    NRS_TRY(c->ensure(d_pos, sizeof(float) * 3 * (size_t)n_points));
    NRS_TRY(c->ensure(d_mind, sizeof(float) * (size_t)n_points));
    NRS_TRY(c->ensure(d_out, sizeof(int) * ((size_t)n_nodes + 1)));
    NRS_HIP(c, hipMemcpyAsync(d_pos.p, pos, sizeof(float) * 3 * (size_t)n_points, hipMemcpyHostToDevice, c->stream));
    if (eligible) {
        NRS_TRY(c->ensure(d_el, (size_t)n_points));
        NRS_HIP(c, hipMemcpyAsync(d_el.p, eligible, (size_t)n_points, hipMemcpyHostToDevice, c->stream));
    }
    hipLaunchKernelGGL(k_fps, dim3(1), dim3(FPS_BLK), 0, c->stream, n_points, d_pos.as<float>(), eligible ? d_el.as<uint8_t>() : nullptr,
                       n_nodes, d_mind.as<float>(), d_out.as<int>(), d_out.as<int>() + n_nodes);
    NRS_HIP(c, hipGetLastError());
    std::vector<int> h((size_t)n_nodes + 1);
    NRS_HIP(c, hipMemcpyAsync(h.data(), d_out.p, sizeof(int) * h.size(), hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    if (h[n_nodes] < n_nodes) return c->fail(NRS_ERR_INVALID, "nrs_skin_select_nodes: %d nodes wanted, %d eligible points", n_nodes, h[n_nodes]);
    for (int k = 0; k < n_nodes; ++k) node_ids[k] = h[k];
    return NRS_OK;
}
