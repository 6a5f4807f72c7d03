// Device-side problem construction for plain deformable-BA windows (part of nrs_engine.hip; see nrs_engine_setup.hpp for the
// host form it reproduces BIT FOR BIT -- tests/test_gpu_devpack.py compares checksums of every packed array).
//
// LocalDeformableBundleAdjustment is called with a NEW window at every keyframe (reference modules/mapping/mapping.cc:57), so
// what a drop-in caller pays is the one-shot nrs_dba_solve: with the packing on a few host threads that was 69-90 ms at C2 for
// a 2.7 ms solve.  Here the caller's arrays (OPT:927-1137's edges as nrs_dba_build_edges leaves them) are uploaded as they are
// and everything else happens on the device, in the order and with the tie-breaks of the host code:
//   row layout      Morton code of every vertex (same fp64 expression), segmented radix sort per keyframe (stable: ties by vertex
//                   index), then inside every 128-row tile a stable sort by (damper, spring) incidence counts
//   sliced ELL      incidence counts per row (integer atomics), slice widths, scans; the k-th incidence of a row is the k-th in
//                   edge order: radix sort of (row, edge sequence) keys
//   halo lists      one workgroup per tile: LDS hash set of the rows referenced outside the tile, spring-referenced rows first,
//                   each part ascending (bitonic sort in LDS), tile-local ids by binary search, headers written in final form
//   chi2 edge lists edges ordered by the row that counts them (stable): radix sort of (row, edge) keys
// The host keeps what it needs to drive the solve: the vertex -> row map (download), the tile classes (from the halo sizes).
// Conditions (otherwise engine_create takes the host path): a plain BA window (nothing fixed, no masks, offsets or unary
// dampers, every damper with four vertices, springs without kernel) of >= 2 keyframes and >= 2048 padded rows -- the two-kernel
// path (>= 32768 rows, T = 2) and the fused one (T = 8) alike -- no communicator, halos that fit the LDS budget, at most 2048 halo
// rows per tile.  NRS_HOST_PACK=1 forces the host path, and so does every A/B switch that path honours (devpack_eligible).
#pragma once
#include <rocprim/rocprim.hpp>

namespace nrs {

constexpr int DP_HCAP = 2048, DP_HASH = 4096;

__device__ inline uint64_t dp_spread(uint64_t v) {                  // 21 bits -> every third bit (as engine_create)
    v &= 0x1fffff;
    v = (v | v << 32) & 0x1f00000000ffffULL;
    v = (v | v << 16) & 0x1f0000ff0000ffULL;
    v = (v | v << 8) & 0x100f00f00f00f00fULL;
    v = (v | v << 4) & 0x10c30c30c30c30c3ULL;
    v = (v | v << 2) & 0x1249249249249249ULL;
    return v;
}

// min / max of the vertex positions (fp64 of the caller's floats): one workgroup, fixed order (min / max are order-free)
__global__ __launch_bounds__(1024) void k_dp_minmax(int M, const double* __restrict__ xyz, double* out /*6*/) {
    __shared__ double lo[3][1024], hi[3][1024];
    const int tid = threadIdx.x;
    double l[3] = {1e300, 1e300, 1e300}, h[3] = {-1e300, -1e300, -1e300};
    for (int v = tid; v < M; v += 1024)
        for (int a = 0; a < 3; ++a) { const double p = xyz[3 * (size_t)v + a]; l[a] = fmin(l[a], p); h[a] = fmax(h[a], p); }
    for (int a = 0; a < 3; ++a) { lo[a][tid] = l[a]; hi[a][tid] = h[a]; }
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s)
            for (int a = 0; a < 3; ++a) { lo[a][tid] = fmin(lo[a][tid], lo[a][tid + s]); hi[a][tid] = fmax(hi[a][tid], hi[a][tid + s]); }
        __syncthreads();
    }
    if (tid < 3) { out[tid] = lo[tid][0]; out[3 + tid] = hi[tid][0]; }
}

__global__ void k_dp_morton(int M, const double* __restrict__ xyz, const double* __restrict__ mm, int morton, uint64_t* code, int* val) {
#pragma clang fp contract(off)
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= M) return;
    uint64_t c = 0;
    if (morton)
        for (int a = 0; a < 3; ++a) {
            const double ext = mm[3 + a] - mm[a];
            const double f = ext > 0 ? (xyz[3 * (size_t)v + a] - mm[a]) / ext : 0.0;
            c |= dp_spread((uint64_t)(f * 2097151.0)) << a;
        }
    code[v] = c;
    val[v] = v;
}

// vertex -> row after the per-keyframe sort: sorted position i of keyframe k -> row pose_grp_ptr[k] * 256 + (i - pose_ptr[k])
__global__ void k_dp_rows0(int M, const int* __restrict__ sorted_v, const int* __restrict__ lm_kf, const int* __restrict__ pose_ptr,
                           const int* __restrict__ pose_grp_ptr, int* vrow, int* row_v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const int v = sorted_v[i], k = lm_kf[v];
    const int r = pose_grp_ptr[k] * ROW_ALIGN + (i - pose_ptr[k]);
    vrow[v] = r;
    row_v[r] = v;
}

// incidence counts per VERTEX (tile sort keys)
__global__ void k_dp_vcounts(int n_sp, const int* __restrict__ sp_ij, int n_dm, const int* __restrict__ dm_idx, int* cs, int* cd) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * (int64_t)n_sp) atomicAdd(&cs[sp_ij[i]], 1);
    if (i < 4 * (int64_t)n_dm) atomicAdd(&cd[dm_idx[i]], 1);
}

// tile sort key of a row: (damper count desc, spring count desc), padding rows last; the segmented sort is stable
__global__ void k_dp_tilekeys(int n_rows, const int* __restrict__ row_v, const int* __restrict__ cs, const int* __restrict__ cd, uint32_t* key, int* val) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const int v = row_v[r];
    key[r] = v < 0 ? 0xFFFFFFFFu : (((uint32_t)(0x7FFF - min(cd[v], 0x7FFF)) << 16) | (uint32_t)(0xFFFF - min(cs[v], 0xFFFF)));
    val[r] = v;
}
__global__ void k_dp_rows1(int n_rows, const int* __restrict__ sorted_v, int* vrow) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const int v = sorted_v[r];
    if (v >= 0) vrow[v] = r;
}

// rows of every incidence + counts per row + sort keys (row << 32 | sequence number in the host's fill order)
__global__ void k_dp_inc(int n_sp, const int* __restrict__ sp_ij, int n_dm, const int* __restrict__ dm_idx, const int* __restrict__ vrow,
                         int* cnt_s, int* cnt_d, uint64_t* key_s, uint64_t* key_d) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * (int64_t)n_sp) {
        const int r = vrow[sp_ij[i]];
        atomicAdd(&cnt_s[r], 1);
        key_s[i] = ((uint64_t)(uint32_t)r << 32) | (uint64_t)(uint32_t)i;
    }
    if (i < 4 * (int64_t)n_dm) {
        const int r = vrow[dm_idx[i]];
        atomicAdd(&cnt_d[r], 1);
        key_d[i] = ((uint64_t)(uint32_t)r << 32) | (uint64_t)(uint32_t)i;
    }
}

// slice widths (x 64): a slice = 64 / T rows, T lanes per row
__global__ void k_dp_widths(int n_slices, int T, const int* __restrict__ cnt_s, const int* __restrict__ cnt_d, int* ws, int* wd) {
    const int sl = blockIdx.x * blockDim.x + threadIdx.x;
    if (sl >= n_slices) return;
    const int Rw = 64 / T;
    int a = 0, b = 0;
    for (int r = 0; r < Rw; ++r) { a = max(a, (cnt_s[sl * Rw + r] + T - 1) / T); b = max(b, (cnt_d[sl * Rw + r] + T - 1) / T); }
    ws[sl] = a * 64;
    wd[sl] = b * 64;
}

__device__ inline size_t dp_pos_of(const int* ptr, int T, int row, int k) {  // packed position of the k-th incidence of a row
    const int Rw = 64 / T, sl = row / Rw, r = row - sl * Rw;
    return (size_t)ptr[sl] + (size_t)(k / T) * 64 + (size_t)r * T + (size_t)(k % T);
}

// fill: sorted incidence i of a row -> its sliced-ELL slot (global neighbour rows; the halo pass turns them into tile-local ids)
__global__ void k_dp_fill_s(int64_t n, int T, const uint64_t* __restrict__ key, const int* __restrict__ row_start, const int* __restrict__ ss_ptr,
                            const int* __restrict__ sp_ij, const float* __restrict__ sp_d0, const int* __restrict__ vrow,
                            int* S_other, float* S_d0, uint8_t* S_side) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int row = (int)(key[i] >> 32);
    const uint32_t seq = (uint32_t)key[i];
    const size_t p = dp_pos_of(ss_ptr, T, row, (int)(i - row_start[row]));
    S_other[p] = vrow[sp_ij[seq ^ 1u]];
    S_d0[p] = sp_d0[seq >> 1];
    S_side[p] = (uint8_t)(1 + (seq & 1u));                           // 1: first endpoint (counts the edge's chi2), 2: second
}
__global__ void k_dp_fill_d(int64_t n, int T, const uint64_t* __restrict__ key, const int* __restrict__ row_start, const int* __restrict__ sd_ptr,
                            const int* __restrict__ dm_idx, const float* __restrict__ dm_w, const int* __restrict__ vrow,
                            int* D_o, float* D_w, int8_t* D_role) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int row = (int)(key[i] >> 32);
    const uint32_t seq = (uint32_t)key[i];
    const int role = (int)(seq & 3u);
    const size_t q = seq >> 2, p = dp_pos_of(sd_ptr, T, row, (int)(i - row_start[row]));
    int z = 0;
    for (int k = 0; k < 4; ++k)
        if (k != role) D_o[3 * p + z++] = vrow[dm_idx[4 * q + k]];
    D_w[p] = dm_w[q];
    D_role[p] = (int8_t)role;
}

// ---- halo of a tile: the rows its incidences reference outside its own 128 rows.  PASS 0 counts (hs, ns), PASS 1 writes
// the lists and the final incidence headers.  LDS: open-addressing hash set (row, seen-by-a-spring flag), then a bitonic sort
// of (damper-only << 31 | row).
template <int PASS>
__global__ __launch_bounds__(256) void k_dp_halo(int tile_rows, int T, uint16_t* row_tp16, const int* __restrict__ ss_ptr, const int* __restrict__ sd_ptr, const int* __restrict__ S_other,
                                                 const uint8_t* __restrict__ S_side, const int* __restrict__ D_o, const int8_t* __restrict__ D_role,
                                                 int* hs, int* hns, const int* __restrict__ halo_ptr, int* halo_rows, uint32_t* s_om, uint2* d_hdr,
                                                 int* overflow) {
    __shared__ uint32_t hk[DP_HASH];                                 // row + 1 (0 = empty)
    __shared__ uint32_t hf[DP_HASH];                                 // 1 = referenced by a spring
    __shared__ uint32_t keys[DP_HCAP];
    __shared__ int cnt, cnt_s;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int row0 = b * tile_rows, row1 = row0 + tile_rows;
    for (int i = tid; i < DP_HASH; i += 256) { hk[i] = 0; hf[i] = 0; }
    if (tid == 0) { cnt = 0; cnt_s = 0; }
    __syncthreads();
    const int s0 = ss_ptr[b * 4], s1 = ss_ptr[b * 4 + 4], d0 = sd_ptr[b * 4], d1 = sd_ptr[b * 4 + 4];
    auto insert = [&](int o, uint32_t spring) {
        if (o < 0 || (o >= row0 && o < row1)) return;
        uint32_t h = ((uint32_t)o * 2654435761u) >> 20;              // 12 bits
        for (int probe = 0; probe < DP_HASH; ++probe, h = (h + 1) & (DP_HASH - 1)) {
            const uint32_t old = atomicCAS(&hk[h], 0u, (uint32_t)o + 1u);
            if (old == 0u || old == (uint32_t)o + 1u) { if (spring) atomicOr(&hf[h], 1u); return; }
        }
        atomicExch(overflow, 1);
    };
    for (int p = s0 + tid; p < s1; p += 256) insert(S_other[p], 1u);
    __syncthreads();
    for (int64_t p = 3 * (int64_t)d0 + tid; p < 3 * (int64_t)d1; p += 256) insert(D_o[p], 0u);
    __syncthreads();
    for (int i = tid; i < DP_HCAP; i += 256) keys[i] = 0xFFFFFFFFu;
    __syncthreads();
    for (int i = tid; i < DP_HASH; i += 256) {
        if (hk[i]) {
            const int j = atomicAdd(&cnt, 1);
            if (hf[i]) atomicAdd(&cnt_s, 1);
            if (j < DP_HCAP) keys[j] = (hf[i] ? 0u : 0x80000000u) | (hk[i] - 1u);
        }
    }
    __syncthreads();
    const int n = cnt, ns = cnt_s;
    if (n > DP_HCAP) { if (tid == 0) atomicExch(overflow, 1); if (PASS == 0 && tid == 0) { hs[b] = n; hns[b] = ns; } return; }
    if (PASS == 0) { if (tid == 0) { hs[b] = n; hns[b] = ns; } return; }
    // bitonic sort of keys[0 .. 2048): spring part first (bit 31 clear), each part ascending by row; padding (0xFFFFFFFF) last
    for (int k = 2; k <= DP_HCAP; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < DP_HCAP; i += 256) {
                const int l = i ^ j;
                if (l > i) {
                    const uint32_t a = keys[i], c = keys[l];
                    const bool up = (i & k) == 0;
                    if ((a > c) == up) { keys[i] = c; keys[l] = a; }
                }
            }
            __syncthreads();
        }
    const int hb = halo_ptr[b];
    for (int i = tid; i < n; i += 256) halo_rows[hb + i] = (int)(keys[i] & 0x7FFFFFFFu);
    auto loc = [&](int o) -> uint32_t {                              // tile-local id as u16 (REC_NONE: no neighbour)
        if (o < 0) return (uint32_t)REC_NONE;
        if (o >= row0 && o < row1) return (uint32_t)(o - row0);
        // spring part [0, ns), damper-only part [ns, n): binary search in both
        int lo = 0, hi = ns;
        while (lo < hi) { const int m = (lo + hi) >> 1; if ((keys[m] & 0x7FFFFFFFu) < (uint32_t)o) lo = m + 1; else hi = m; }
        if (lo < ns && (keys[lo] & 0x7FFFFFFFu) == (uint32_t)o) return (uint32_t)(tile_rows + lo);
        lo = ns; hi = n;
        while (lo < hi) { const int m = (lo + hi) >> 1; if ((keys[m] & 0x7FFFFFFFu) < (uint32_t)o) lo = m + 1; else hi = m; }
        return (uint32_t)(tile_rows + lo);
    };
    // springs: {other u16 | meta u16 << 16}, meta = SR_ACTIVE | SR_COUNT on the edge's first endpoint; padding: other = REC_NONE, meta 0
    for (int p = s0 + tid; p < s1; p += 256) {
        const int o = S_other[p];
        const uint32_t side = S_side[p];
        const uint32_t m16 = o < 0 ? 0u : (uint32_t)(SR_ACTIVE | (side == 1 ? SR_COUNT : 0));
        s_om[p] = loc(o) | (m16 << 16);
    }
    // dampers: the three others in canonical order (engine_create: perm), meta = role | DM_ACTIVE | DM_COUNT on role 0
    for (int p = d0 + tid; p < d1; p += 256) {
        const int role = D_role[p];
        if (role < 0) { d_hdr[p] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu); continue; }
        const int pm0 = role < 2 ? 0 : 2, pm1 = (role == 0) ? 1 : (role == 1 ? 2 : (role == 2 ? 0 : 1)), pm2 = (role == 0) ? 2 : (role == 1 ? 1 : (role == 2 ? 1 : 0));
        const uint32_t l0 = loc(D_o[3 * (size_t)p + pm0]), l1 = loc(D_o[3 * (size_t)p + pm1]), l2 = loc(D_o[3 * (size_t)p + pm2]);
        const uint32_t m16 = (uint32_t)(role | DM_ACTIVE | (role == 0 ? DM_COUNT : 0));
        d_hdr[p] = make_uint2(l0 | (l1 << 16), l2 | (m16 << 16));
        // the row's own temporal partner (l1): next for roles 1c / 2c, previous for 1n / 2n -- the same for all of its dampers
        const int sl = b * 4 + (p >= sd_ptr[b * 4 + 1]) + (p >= sd_ptr[b * 4 + 2]) + (p >= sd_ptr[b * 4 + 3]);
        const int row = sl * (64 / T) + ((p - sd_ptr[sl]) & 63) / T;
        row_tp16[2 * (size_t)row + (role < 2 ? 0 : 1)] = (uint16_t)l1;
    }
    __syncthreads();
    for (int p = d0 + tid; p < d1; p += 256) {                       // ... which is verified, not assumed
        const int role = D_role[p];
        if (role < 0) continue;
        const int sl = b * 4 + (p >= sd_ptr[b * 4 + 1]) + (p >= sd_ptr[b * 4 + 2]) + (p >= sd_ptr[b * 4 + 3]);
        const int row = sl * (64 / T) + ((p - sd_ptr[sl]) & 63) / T;
        if (row_tp16[2 * (size_t)row + (role < 2 ? 0 : 1)] != (uint16_t)(d_hdr[p].x >> 16)) atomicExch(overflow + 1, 1);
    }
}

// chi2 edge lists: keys (counting row << 32 | edge), then the records in sorted order
__global__ void k_dp_eckeys(int n_sp, const int* __restrict__ sp_ij, int n_dm, const int* __restrict__ dm_idx, const int* __restrict__ vrow,
                            uint64_t* ks, uint64_t* kd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_sp) ks[i] = ((uint64_t)(uint32_t)vrow[sp_ij[2 * (size_t)i]] << 32) | (uint32_t)i;
    if (i < n_dm) kd[i] = ((uint64_t)(uint32_t)vrow[dm_idx[4 * (size_t)i]] << 32) | (uint32_t)i;
}
__global__ void k_dp_ecfill(int n_sp, const uint64_t* __restrict__ ks, const int* __restrict__ sp_ij, const float* __restrict__ sp_d0, int n_dm,
                            const uint64_t* __restrict__ kd, const int* __restrict__ dm_idx, const float* __restrict__ dm_w, const int* __restrict__ vrow,
                            EcSpring* ec_sp, EcDamper* ec_dm, float* ec_w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_sp) {
        const uint32_t q = (uint32_t)ks[i];
        ec_sp[i] = EcSpring{vrow[sp_ij[2 * (size_t)q]], vrow[sp_ij[2 * (size_t)q + 1]], sp_d0[q], 0};
    }
    if (i < n_dm) {
        const uint32_t q = (uint32_t)kd[i];
        EcDamper e;
        for (int k = 0; k < 4; ++k) e.r[k] = vrow[dm_idx[4 * (size_t)q + k]];
        ec_dm[i] = e;
        ec_w[i] = dm_w[q];
    }
}

// per-row data from per-vertex data (padding rows: fixed, no observation, zeros)
__global__ void k_dp_rowdata(int n_rows, const int* __restrict__ row_v, const double* __restrict__ xyz, const float* __restrict__ uv_in,
                             uint8_t* rflag, float* uv, double* xl) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const int v = row_v[r];
    rflag[r] = v < 0 ? (uint8_t)RF_FIXED : (uint8_t)(RF_OBS | RF_REPROJ_ACTIVE);
    uv[2 * (size_t)r] = v < 0 ? 0.f : uv_in[2 * (size_t)v];
    uv[2 * (size_t)r + 1] = v < 0 ? 0.f : uv_in[2 * (size_t)v + 1];
    for (int a = 0; a < 3; ++a) xl[3 * (size_t)r + a] = v < 0 ? 0.0 : xyz[3 * (size_t)v + a];
}
__global__ void k_dp_rowcnt(int n_rows, const int* __restrict__ cnt_s, const int* __restrict__ cnt_d, uint32_t* out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_rows) out[r] = (uint32_t)cnt_s[r] | ((uint32_t)cnt_d[r] << 16);
}
__global__ void k_dp_rowv(int M, const int* __restrict__ vrow, int* row_v) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < M) row_v[vrow[v]] = v;
}

// fused single-launch path: the first BLK halo rows of every tile at a fixed stride (no pointer chase in k_pcg_fused)
// (plain two-kernel path: the first HALO_FIX rows, -1 behind the list's end -- stage_rows<true>)
__global__ void k_dp_halofix(int n_tiles, const int* __restrict__ halo_ptr, const int* __restrict__ halo_rows, int* halo_fix, int stride, int pad) {
    const int b = blockIdx.x;
    if (b >= n_tiles) return;
    const int hb = halo_ptr[b], hn = halo_ptr[b + 1] - hb;
    for (int i = threadIdx.x; i < stride; i += blockDim.x) halo_fix[(size_t)b * stride + i] = i < hn ? halo_rows[hb + i] : pad;
}

// bump allocator over the context's pack scratch
struct DpScratch {
    char* base;
    size_t off = 0, cap;
    template <class Tp> Tp* get(size_t n) {
        Tp* p = reinterpret_cast<Tp*>(base + off);
        off += ((n * sizeof(Tp) + 255) / 256) * 256 + 256;
        return p;
    }
};

// ---- edge construction of LocalDeformableBundleAdjustment on the device (OPT:927-1137; the host form is nrs_dba_build_edges,
// csrc/nrs_host_build.cpp, whose output this reproduces index for index).  One thread per landmark l = (keyframe k, map point p):
// it walks p's ordered neighbour list as the reference does (stop after more than 10 regularisers or at the first BAD
// connection, OPT:1033-1050; a duplicate counts as a regulariser too, so the walked prefix of a point does not depend on any
// other point).  The reference's per-keyframe hash sets (SpatialPoint / TemporalPoint, OPT:980-981) keep the FIRST of the two
// walks that propose a pair: the pair {p, o} is a duplicate here iff o comes earlier in the keyframe and o's own walk reaches p.
struct WinDev {
    int n_kf, n_points;
    const int *kf_rowptr, *kf_pt, *lm_k;                             // lm_k[l] = keyframe of landmark l
    const int *nbr_rowptr, *nbr_col, *nbr_status;
    const float *nbr_w, *nbr_d0;
    const int* cur;                                                  // n_kf x n_points: landmark of (k, p) or -1
};

__device__ inline bool win_walk_reaches(const WinDev& W, int k, int o, int p, bool damper) {
    const int* cur = W.cur + (size_t)k * W.n_points;
    const int* nxt = damper ? W.cur + (size_t)(k + 1) * W.n_points : nullptr;
    int n_reg = 0;
    for (int e = W.nbr_rowptr[o]; e < W.nbr_rowptr[o + 1]; ++e) {
        if (n_reg > 10 || W.nbr_status[e] == NRS_GRAPH_BAD) return false;
        const int x = W.nbr_col[e];
        if (cur[x] < 0 || (damper && nxt[x] < 0)) continue;
        if (x == p) return true;
        ++n_reg;
    }
    return false;
}

// EMIT = false: counts per landmark; true: writes at the scanned offsets
template <bool EMIT>
__global__ void k_win_edges(WinDev W, int n_lm, int* cnt_s, int* cnt_d, const int* __restrict__ off_s, const int* __restrict__ off_d,
                            int* sp_ij, float* sp_d0, int* dm_idx, float* dm_w) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n_lm) return;
    const int k = W.lm_k[l], p = W.kf_pt[l];
    const int* cur = W.cur + (size_t)k * W.n_points;
    const int lo = W.nbr_rowptr[p], hi = W.nbr_rowptr[p + 1];
    int ns = 0, nd = 0, n_reg = 0;
    for (int e = lo; e < hi; ++e) {                                  // springs (OPT:1033-1074)
        if (n_reg > 10 || W.nbr_status[e] == NRS_GRAPH_BAD) break;
        const int o = W.nbr_col[e];
        if (cur[o] < 0) continue;
        ++n_reg;
        if (cur[o] < l && win_walk_reaches(W, k, o, p, false)) continue;        // inserted by o's walk already
        if (EMIT) { const int q = off_s[l] + ns; sp_ij[2 * (size_t)q] = l; sp_ij[2 * (size_t)q + 1] = cur[o]; sp_d0[q] = W.nbr_d0[e]; }
        ++ns;
    }
    if (k + 1 < W.n_kf) {                                            // dampers with the next keyframe (OPT:1076-1136)
        const int* nxt = W.cur + (size_t)(k + 1) * W.n_points;
        if (nxt[p] >= 0) {
            n_reg = 0;
            for (int e = lo; e < hi; ++e) {
                if (n_reg > 10 || W.nbr_status[e] == NRS_GRAPH_BAD) break;
                const int o = W.nbr_col[e];
                if (cur[o] < 0 || nxt[o] < 0) continue;
                ++n_reg;
                if (cur[o] < l && win_walk_reaches(W, k, o, p, true)) continue;
                if (EMIT) {
                    const int q = off_d[l] + nd;
                    dm_idx[4 * (size_t)q] = l; dm_idx[4 * (size_t)q + 1] = cur[o]; dm_idx[4 * (size_t)q + 2] = nxt[p]; dm_idx[4 * (size_t)q + 3] = nxt[o];
                    dm_w[q] = W.nbr_w[e];
                }
                ++nd;
            }
        }
    }
    if (!EMIT) { cnt_s[l] = ns; cnt_d[l] = nd; }
}
__global__ void k_win_cur(int n_lm, const int* __restrict__ lm_k, const int* __restrict__ kf_pt, int n_points, int* cur) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l < n_lm) atomicMax(&cur[(size_t)lm_k[l] * n_points + kf_pt[l]], l);   // (a map point listed twice in a keyframe: the last entry wins, as on the host)
}

// Builds the edge lists of a window on the device; the arrays live in the context's third scratch buffer until the next call.
int engine_build_edges_device(nrs_ctx* c, int n_kf, const int* kf_rowptr, const int* kf_pt, const int* lm_kf, int n_points, const int* nbr_rowptr,
                              const int* nbr_col, const float* nbr_w, const float* nbr_d0, const int* nbr_status, DevEdges* out) {
    const int n_lm = kf_rowptr[n_kf], nnz = nbr_rowptr[n_points];
    hipStream_t st = c->stream;
    size_t tb = 0;
    (void)rocprim::exclusive_scan(nullptr, tb, (int*)nullptr, (int*)nullptr, 0, (size_t)n_lm + 1, rocprim::plus<int>(), st);
    int *d_rowptr, *d_pt, *d_k, *d_nrp, *d_col, *d_st, *d_cur, *d_cs, *d_cd, *d_os, *d_od;
    float *d_w, *d_d0;
    void* tmp;
    auto layout = [&](DpScratch& W) {
        d_rowptr = W.get<int>(n_kf + 1); d_pt = W.get<int>(n_lm); d_k = W.get<int>(n_lm);
        d_nrp = W.get<int>(n_points + 1); d_col = W.get<int>(nnz); d_st = W.get<int>(nnz); d_w = W.get<float>(nnz); d_d0 = W.get<float>(nnz);
        d_cur = W.get<int>((size_t)n_kf * n_points);
        d_cs = W.get<int>((size_t)n_lm + 1); d_cd = W.get<int>((size_t)n_lm + 1); d_os = W.get<int>((size_t)n_lm + 1); d_od = W.get<int>((size_t)n_lm + 1);
        tmp = W.get<char>(tb + 256);
    };
    DpScratch dry{nullptr, 0, 0};
    layout(dry);
    NRS_TRY(c->ensure(c->pack_ws3, dry.off + 4096));
    DpScratch W{c->pack_ws3.as<char>(), 0, c->pack_ws3.cap};
    layout(W);
    NRS_HIP(c, hipMemcpyAsync(d_rowptr, kf_rowptr, sizeof(int) * (n_kf + 1), hipMemcpyHostToDevice, st));
    NRS_HIP(c, hipMemcpyAsync(d_pt, kf_pt, sizeof(int) * (size_t)n_lm, hipMemcpyHostToDevice, st));
    NRS_HIP(c, hipMemcpyAsync(d_k, lm_kf, sizeof(int) * (size_t)n_lm, hipMemcpyHostToDevice, st));
    NRS_HIP(c, hipMemcpyAsync(d_nrp, nbr_rowptr, sizeof(int) * ((size_t)n_points + 1), hipMemcpyHostToDevice, st));
    NRS_HIP(c, hipMemcpyAsync(d_col, nbr_col, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice, st));
    NRS_HIP(c, hipMemcpyAsync(d_st, nbr_status, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice, st));
    NRS_HIP(c, hipMemcpyAsync(d_w, nbr_w, sizeof(float) * (size_t)nnz, hipMemcpyHostToDevice, st));
    NRS_HIP(c, hipMemcpyAsync(d_d0, nbr_d0, sizeof(float) * (size_t)nnz, hipMemcpyHostToDevice, st));
    NRS_HIP(c, hipMemsetAsync(d_cur, 0xFF, sizeof(int) * (size_t)n_kf * n_points, st));
    NRS_HIP(c, hipMemsetAsync(d_cs + n_lm, 0, sizeof(int), st));
    NRS_HIP(c, hipMemsetAsync(d_cd + n_lm, 0, sizeof(int), st));
    const dim3 g((unsigned)((n_lm + 255) / 256)), b(256);
    hipLaunchKernelGGL(k_win_cur, g, b, 0, st, n_lm, d_k, d_pt, n_points, d_cur);
    WinDev wd{n_kf, n_points, d_rowptr, d_pt, d_k, d_nrp, d_col, d_st, d_w, d_d0, d_cur};
    hipLaunchKernelGGL((k_win_edges<false>), g, b, 0, st, wd, n_lm, d_cs, d_cd, (const int*)nullptr, (const int*)nullptr, (int*)nullptr, (float*)nullptr, (int*)nullptr, (float*)nullptr);
    size_t t2 = tb + 256;
    NRS_HIP(c, rocprim::exclusive_scan(tmp, t2, d_cs, d_os, 0, (size_t)n_lm + 1, rocprim::plus<int>(), st));
    t2 = tb + 256;
    NRS_HIP(c, rocprim::exclusive_scan(tmp, t2, d_cd, d_od, 0, (size_t)n_lm + 1, rocprim::plus<int>(), st));
    int tot[2] = {0, 0};
    NRS_HIP(c, hipMemcpyAsync(&tot[0], d_os + n_lm, sizeof(int), hipMemcpyDeviceToHost, st));
    NRS_HIP(c, hipMemcpyAsync(&tot[1], d_od + n_lm, sizeof(int), hipMemcpyDeviceToHost, st));
    NRS_HIP(c, hipStreamSynchronize(st));
    auto al = [](size_t x) { return ((x + 255) / 256) * 256 + 256; };
    NRS_TRY(c->ensure(c->pack_ws4, al(8 * (size_t)tot[0]) + al(4 * (size_t)tot[0]) + al(16 * (size_t)tot[1]) + al(4 * (size_t)tot[1]) + 4096));
    DpScratch W4{c->pack_ws4.as<char>(), 0, c->pack_ws4.cap};
    int* sp_ij = W4.get<int>(2 * (size_t)tot[0]); float* sp_d0 = W4.get<float>(tot[0]);
    int* dm_idx = W4.get<int>(4 * (size_t)tot[1]); float* dm_w = W4.get<float>(tot[1]);
    hipLaunchKernelGGL((k_win_edges<true>), g, b, 0, st, wd, n_lm, (int*)nullptr, (int*)nullptr, d_os, d_od, sp_ij, sp_d0, dm_idx, dm_w);
    NRS_HIP(c, hipGetLastError());
    out->n_sp = tot[0]; out->n_dm = tot[1];
    out->sp_ij = sp_ij; out->sp_d0 = sp_d0; out->dm_idx = dm_idx; out->dm_w = dm_w;
    return NRS_OK;
}

static bool devpack_eligible(nrs_ctx* c, const EngineSpec& s, int n_pad_rows) {
    if (c->env("NRS_HOST_PACK") || c->env("NRS_NO_PLAIN") || c->env("NRS_NO_LDS") || c->env("NRS_DFORM") || c->env("NRS_NO_EDGE_CHI") || c->env("NRS_NO_FUSED") ||
        c->env("NRS_SELL_T") || c->env("NRS_FUSED_MAX_ROWS") || c->env("NRS_TILE_CUT_PCT") || c->env("NRS_HIER") || c->env("NRS_NO_ECD"))
        return false;                                                // (test / A-B switches are honoured by the host path)
    if (c->comm || s.X0 || s.n_un || s.sp_active || s.dm_active || s.pose_fixed || s.force_gather || s.n_skin > 0) return false;
    if (s.K < 2 || n_pad_rows < 2048 || s.delta_pos > 0 || s.spring_form != 0 || s.n_dm <= 0 || s.n_sp <= 0) return false;   // (single-frame problems: a2, host)
    if (4 * (int64_t)s.n_dm >= 0xFFFFFFFFLL || (int64_t)n_pad_rows >= 0x7FFFFFFFLL) return false;
    return true;
}

bool engine_device_pack_ok(nrs_ctx* c, const EngineSpec& s) {
    std::vector<int> cnt(s.K, 0);
    for (int i = 0; i < s.M; ++i) cnt[s.lm_pose[i]]++;
    int n_pad = 0;
    for (int k = 0; k < s.K; ++k) n_pad += std::max(1, (cnt[k] + ROW_ALIGN - 1) / ROW_ALIGN) * ROW_ALIGN;
    return devpack_eligible(c, s, n_pad);
}

int engine_edges_to_host(nrs_ctx* c, Engine* e, int* sp_ij, float* sp_d0, int* dm_idx, float* dm_w) {
    if (!e->dev_edges) return c->fail(NRS_ERR_STATE, "the resident problem keeps its edges on the host");
    const Dev& d = e->d;
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    if (sp_ij) NRS_HIP(c, hipMemcpy(sp_ij, e->raw_sp, sizeof(int) * 2 * (size_t)d.n_sp, hipMemcpyDeviceToHost));
    if (sp_d0) NRS_HIP(c, hipMemcpy(sp_d0, e->raw_d0, sizeof(float) * (size_t)d.n_sp, hipMemcpyDeviceToHost));
    if (dm_idx) NRS_HIP(c, hipMemcpy(dm_idx, e->raw_dm, sizeof(int) * 4 * (size_t)d.n_dm, hipMemcpyDeviceToHost));
    if (dm_w) NRS_HIP(c, hipMemcpy(dm_w, e->raw_w, sizeof(float) * (size_t)d.n_dm, hipMemcpyDeviceToHost));
    return NRS_OK;
}

// Returns NRS_OK with *done = true when the engine was built here; *done = false (and NRS_OK) when the window turned out not to
// qualify (a fixed vertex, an incomplete damper, a halo beyond the limits): the caller then runs the host path.
static int engine_create_device(nrs_ctx* c, const EngineSpec& s, Arena* arena, Engine* e, bool* done) {
    *done = false;
    Dev& d = e->d;
    const bool tm = c->env("NRS_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!tm) return;
        (void)hipStreamSynchronize(c->stream);
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[nrs] device pack %-18s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    for (int v = 0; v < s.M; ++v) if (s.rflag[v] != (RF_OBS | RF_REPROJ_ACTIVE)) return NRS_OK;
    if (!s.edges_on_device)                                        // an incomplete damper (-1: absent vertex, valid for engine_create): the device
        for (int64_t q = 0; q < 4 * (int64_t)s.n_dm; ++q)          // kernels index by these values, so such a window takes the host path
            if (s.dm_idx[q] < 0) return NRS_OK;
    const int K = s.K, M = s.M, n_sp = s.n_sp, n_dm = s.n_dm;
    std::vector<int> pose_ptr(K + 1, 0), pose_grp_ptr(K + 1, 0), grp_pose;
    for (int i = 0; i < M; ++i) pose_ptr[s.lm_pose[i] + 1]++;
    for (int k = 0; k < K; ++k) pose_ptr[k + 1] += pose_ptr[k];
    for (int k = 0; k < K; ++k) {
        const int ng = std::max(1, (pose_ptr[k + 1] - pose_ptr[k] + ROW_ALIGN - 1) / ROW_ALIGN);
        pose_grp_ptr[k + 1] = pose_grp_ptr[k] + ng;
        for (int g = 0; g < ng; ++g) grp_pose.push_back(k);
    }
    memset(&d, 0, sizeof(d));
    const int n_pad = pose_grp_ptr[K] * ROW_ALIGN;
    const int T = n_pad >= 32768 ? 2 : 8;                          // lanes per row, as engine_create
    d.T = T; d.K = K; d.M = M; d.n_sp = n_sp; d.n_dm = n_dm; d.n_un = 0;
    d.cam = s.cam;
    d.info_reproj = s.info_reproj; d.delta_reproj = s.delta_reproj; d.info_pos = s.info_pos; d.delta_pos = s.delta_pos;
    d.info_spatial = s.info_spatial; d.delta_spatial = s.delta_spatial; d.k_spring = s.k_spring; d.spring_form = s.spring_form;
    d.n_groups = pose_grp_ptr[K];
    d.n_rows = d.n_groups * ROW_ALIGN;
    d.tile_rows = BLK / T;
    d.n_regblk = d.n_rows / d.tile_rows;
    d.n_vecblk = d.n_rows / BLK;
    const int n_rows = d.n_rows, n_slices = n_rows / (64 / T), n_tiles = d.n_regblk;
    const int64_t ni_s = 2 * (int64_t)n_sp, ni_d = 4 * (int64_t)n_dm;
    // ---- scratch: raw inputs + intermediates (sized from the inputs; the packed arrays themselves go into the arena later)
    size_t tmp_bytes = 0;
    {
        size_t b1 = 0, b2 = 0, b3 = 0, b4 = 0;
        (void)rocprim::radix_sort_keys(nullptr, b1, (uint64_t*)nullptr, (uint64_t*)nullptr, (size_t)std::max(ni_s, ni_d), 0, 64, c->stream);
        (void)rocprim::segmented_radix_sort_pairs(nullptr, b2, (uint64_t*)nullptr, (uint64_t*)nullptr, (int*)nullptr, (int*)nullptr, (size_t)M, (unsigned)K,
                                                  (int*)nullptr, (int*)nullptr, 0, 64, c->stream);
        (void)rocprim::segmented_radix_sort_pairs(nullptr, b3, (uint32_t*)nullptr, (uint32_t*)nullptr, (int*)nullptr, (int*)nullptr, (size_t)n_rows,
                                                  (unsigned)n_tiles, (int*)nullptr, (int*)nullptr, 0, 32, c->stream);
        (void)rocprim::exclusive_scan(nullptr, b4, (int*)nullptr, (int*)nullptr, 0, (size_t)n_rows + 1, rocprim::plus<int>(), c->stream);
        tmp_bytes = std::max(std::max(b1, b2), std::max(b3, b4)) + 1024;
    }
    auto al = [](size_t b) { return ((b + 255) / 256) * 256 + 256; };
    double* r_x; int* r_kf; float* r_uv; int* r_sp; float* r_d0; int* r_dm; float* r_w;
    uint64_t *code_a, *code_b, *key_s, *key_s2, *key_d, *key_d2;
    int *val_a, *val_b, *cs, *cd, *d_pose_ptr, *d_pgp, *tile_off, *vrow, *row_v, *tv_a, *tv_b, *cnt_s, *cnt_d, *rs_s, *rs_d, *ws, *wd, *d_ss, *d_sd, *hs, *hns, *d_flag;
    uint32_t *tk_a, *tk_b;
    void* tmp;
    double* mm;
    auto layout = [&](DpScratch& W) {
        r_x = W.get<double>(3 * (size_t)M); r_kf = W.get<int>(M); r_uv = W.get<float>(2 * (size_t)M);
        r_sp = W.get<int>(2 * (size_t)n_sp); r_d0 = W.get<float>(n_sp); r_dm = W.get<int>(4 * (size_t)n_dm); r_w = W.get<float>(n_dm);
        code_a = W.get<uint64_t>(M); code_b = W.get<uint64_t>(M);
        val_a = W.get<int>(M); val_b = W.get<int>(M); cs = W.get<int>(M); cd = W.get<int>(M);
        d_pose_ptr = W.get<int>(K + 1); d_pgp = W.get<int>(K + 1); tile_off = W.get<int>(n_tiles + 1);
        vrow = W.get<int>(M); row_v = W.get<int>(n_rows);
        tk_a = W.get<uint32_t>(n_rows); tk_b = W.get<uint32_t>(n_rows); tv_a = W.get<int>(n_rows); tv_b = W.get<int>(n_rows);
        cnt_s = W.get<int>((size_t)n_rows + 1); cnt_d = W.get<int>((size_t)n_rows + 1); rs_s = W.get<int>((size_t)n_rows + 1); rs_d = W.get<int>((size_t)n_rows + 1);
        key_s = W.get<uint64_t>(ni_s); key_s2 = W.get<uint64_t>(ni_s); key_d = W.get<uint64_t>(ni_d); key_d2 = W.get<uint64_t>(ni_d);
        ws = W.get<int>((size_t)n_slices + 1); wd = W.get<int>((size_t)n_slices + 1); d_ss = W.get<int>((size_t)n_slices + 1); d_sd = W.get<int>((size_t)n_slices + 1);
        tmp = W.get<char>(tmp_bytes);
        mm = W.get<double>(8);
        hs = W.get<int>(n_tiles); hns = W.get<int>(n_tiles); d_flag = W.get<int>(16);
    };
    {
        DpScratch dry{nullptr, 0, 0};
        layout(dry);
        NRS_TRY(c->ensure(c->pack_ws, dry.off + 4096));
    }
    DpScratch W{c->pack_ws.as<char>(), 0, c->pack_ws.cap};
    layout(W);
    // ---- uploads of the caller's arrays, as they are
    hipStream_t st = c->stream;
    NRS_HIP(c, hipMemcpyAsync(r_x, s.x, sizeof(double) * 3 * (size_t)M, hipMemcpyHostToDevice, st));
    NRS_HIP(c, hipMemcpyAsync(r_kf, s.lm_pose, sizeof(int) * (size_t)M, hipMemcpyHostToDevice, st));
    NRS_HIP(c, hipMemcpyAsync(r_uv, s.uv, sizeof(float) * 2 * (size_t)M, hipMemcpyHostToDevice, st));
    const hipMemcpyKind ek = s.edges_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;      // (nrs_dba_solve_window builds them there)
    NRS_HIP(c, hipMemcpyAsync(r_sp, s.sp_ij, sizeof(int) * 2 * (size_t)n_sp, ek, st));
    NRS_HIP(c, hipMemcpyAsync(r_d0, s.sp_d0, sizeof(float) * (size_t)n_sp, ek, st));
    NRS_HIP(c, hipMemcpyAsync(r_dm, s.dm_idx, sizeof(int) * 4 * (size_t)n_dm, ek, st));
    NRS_HIP(c, hipMemcpyAsync(r_w, s.dm_w, sizeof(float) * (size_t)n_dm, ek, st));
    NRS_HIP(c, hipMemcpyAsync(d_pose_ptr, pose_ptr.data(), sizeof(int) * (K + 1), hipMemcpyHostToDevice, st));
    NRS_HIP(c, hipMemcpyAsync(d_pgp, pose_grp_ptr.data(), sizeof(int) * (K + 1), hipMemcpyHostToDevice, st));
    {
        std::vector<int> to(n_tiles + 1);
        for (int i = 0; i <= n_tiles; ++i) to[i] = i * d.tile_rows;
        NRS_HIP(c, hipMemcpyAsync(tile_off, to.data(), sizeof(int) * (n_tiles + 1), hipMemcpyHostToDevice, st));
        NRS_HIP(c, hipStreamSynchronize(st));                        // (`to` dies here)
    }
    mark("uploads");
    auto nb = [](int64_t n) { return dim3((unsigned)((n + 255) / 256)); };
    // ---- row layout
    hipLaunchKernelGGL(k_dp_minmax, dim3(1), dim3(1024), 0, st, M, r_x, mm);
    hipLaunchKernelGGL(k_dp_morton, nb(M), dim3(256), 0, st, M, r_x, mm, c->env("NRS_NO_MORTON") ? 0 : 1, code_a, val_a);
    size_t tb = tmp_bytes;
    NRS_HIP(c, rocprim::segmented_radix_sort_pairs(tmp, tb, code_a, code_b, val_a, val_b, (size_t)M, (unsigned)K, d_pose_ptr, d_pose_ptr + 1, 0, 64, st));
    NRS_HIP(c, hipMemsetAsync(row_v, 0xFF, sizeof(int) * (size_t)n_rows, st));
    hipLaunchKernelGGL(k_dp_rows0, nb(M), dim3(256), 0, st, M, val_b, r_kf, d_pose_ptr, d_pgp, vrow, row_v);
    if (!c->env("NRS_NO_TILE_SORT")) {
        NRS_HIP(c, hipMemsetAsync(cs, 0, sizeof(int) * (size_t)M, st));
        NRS_HIP(c, hipMemsetAsync(cd, 0, sizeof(int) * (size_t)M, st));
        hipLaunchKernelGGL(k_dp_vcounts, nb(std::max(ni_s, ni_d)), dim3(256), 0, st, n_sp, r_sp, n_dm, r_dm, cs, cd);
        hipLaunchKernelGGL(k_dp_tilekeys, nb(n_rows), dim3(256), 0, st, n_rows, row_v, cs, cd, tk_a, tv_a);
        tb = tmp_bytes;
        NRS_HIP(c, rocprim::segmented_radix_sort_pairs(tmp, tb, tk_a, tk_b, tv_a, tv_b, (size_t)n_rows, (unsigned)n_tiles, tile_off, tile_off + 1, 0, 32, st));
        hipLaunchKernelGGL(k_dp_rows1, nb(n_rows), dim3(256), 0, st, n_rows, tv_b, vrow);
        NRS_HIP(c, hipMemsetAsync(row_v, 0xFF, sizeof(int) * (size_t)n_rows, st));
        hipLaunchKernelGGL(k_dp_rowv, nb(M), dim3(256), 0, st, M, vrow, row_v);
    }
    mark("row layout");
    // ---- incidence counts, slice widths, offsets
    NRS_HIP(c, hipMemsetAsync(cnt_s, 0, sizeof(int) * ((size_t)n_rows + 1), st));
    NRS_HIP(c, hipMemsetAsync(cnt_d, 0, sizeof(int) * ((size_t)n_rows + 1), st));
    hipLaunchKernelGGL(k_dp_inc, nb(std::max(ni_s, ni_d)), dim3(256), 0, st, n_sp, r_sp, n_dm, r_dm, vrow, cnt_s, cnt_d, key_s, key_d);
    hipLaunchKernelGGL(k_dp_widths, nb(n_slices), dim3(256), 0, st, n_slices, T, cnt_s, cnt_d, ws, wd);
    NRS_HIP(c, hipMemsetAsync(ws + n_slices, 0, sizeof(int), st));
    NRS_HIP(c, hipMemsetAsync(wd + n_slices, 0, sizeof(int), st));
    tb = tmp_bytes; NRS_HIP(c, rocprim::exclusive_scan(tmp, tb, ws, d_ss, 0, (size_t)n_slices + 1, rocprim::plus<int>(), st));
    tb = tmp_bytes; NRS_HIP(c, rocprim::exclusive_scan(tmp, tb, wd, d_sd, 0, (size_t)n_slices + 1, rocprim::plus<int>(), st));
    tb = tmp_bytes; NRS_HIP(c, rocprim::exclusive_scan(tmp, tb, cnt_s, rs_s, 0, (size_t)n_rows + 1, rocprim::plus<int>(), st));
    tb = tmp_bytes; NRS_HIP(c, rocprim::exclusive_scan(tmp, tb, cnt_d, rs_d, 0, (size_t)n_rows + 1, rocprim::plus<int>(), st));
    int h_nnz[2] = {0, 0};
    NRS_HIP(c, hipMemcpyAsync(&h_nnz[0], d_ss + n_slices, sizeof(int), hipMemcpyDeviceToHost, st));
    NRS_HIP(c, hipMemcpyAsync(&h_nnz[1], d_sd + n_slices, sizeof(int), hipMemcpyDeviceToHost, st));
    // (the sorts run while the host waits for the two totals)
    const int row_bits = 32 - __builtin_clz((unsigned)std::max(1, n_rows - 1));
    tb = tmp_bytes; NRS_HIP(c, rocprim::radix_sort_keys(tmp, tb, key_s, key_s2, (size_t)ni_s, 0, 32 + row_bits, st));
    tb = tmp_bytes; NRS_HIP(c, rocprim::radix_sort_keys(tmp, tb, key_d, key_d2, (size_t)ni_d, 0, 32 + row_bits, st));
    NRS_HIP(c, hipStreamSynchronize(st));
    const size_t nnz_s = (size_t)h_nnz[0], nnz_d = (size_t)h_nnz[1];
    d.ss_nnz = (int)nnz_s; d.sd_nnz = (int)nnz_d;
    mark("counts + sorts");
    // ---- second scratch: sliced-ELL intermediates with GLOBAL neighbour rows
    const size_t need2 = 2 * al(4 * nnz_s) + al(nnz_s) + al(12 * nnz_d) + al(4 * nnz_d) + al(nnz_d) + 2 * al(8 * (size_t)n_sp) + 2 * al(8 * (size_t)n_dm) + (1 << 16);
    NRS_TRY(c->ensure(c->pack_ws2, need2));
    DpScratch W2{c->pack_ws2.as<char>(), 0, c->pack_ws2.cap};
    int* S_other = W2.get<int>(nnz_s);
    uint8_t* S_side = W2.get<uint8_t>(nnz_s);
    int* D_o = W2.get<int>(3 * nnz_d);
    int8_t* D_role = W2.get<int8_t>(nnz_d);
    uint64_t* ek_s = W2.get<uint64_t>(n_sp); uint64_t* ek_s2 = W2.get<uint64_t>(n_sp);
    uint64_t* ek_d = W2.get<uint64_t>(n_dm); uint64_t* ek_d2 = W2.get<uint64_t>(n_dm);
    float* t_d0 = W2.get<float>(nnz_s);
    float* t_w = W2.get<float>(nnz_d);
    if (W2.off > W2.cap) return c->fail(NRS_ERR_ALLOC, "device pack: scratch under-sized");
    // ---- halo sizes (pass 0 needs S_other / D_o: fill them into scratch first; S_d0 / D_w go straight into the arena later,
    // so the fill kernels run twice as cheaply as once with a staging copy: here only the ids)
    NRS_HIP(c, hipMemsetAsync(S_other, 0xFF, sizeof(int) * nnz_s, st));
    NRS_HIP(c, hipMemsetAsync(S_side, 0, nnz_s, st));
    NRS_HIP(c, hipMemsetAsync(D_o, 0xFF, sizeof(int) * 3 * nnz_d, st));
    NRS_HIP(c, hipMemsetAsync(D_role, 0xFF, nnz_d, st));
    NRS_HIP(c, hipMemsetAsync(d_flag, 0, sizeof(int) * 16, st));
    // (S_d0 / D_w are staged too: the arena is carved only once the halo sizes are known)
    NRS_HIP(c, hipMemsetAsync(t_d0, 0, sizeof(float) * nnz_s, st));
    NRS_HIP(c, hipMemsetAsync(t_w, 0, sizeof(float) * nnz_d, st));
    hipLaunchKernelGGL(k_dp_fill_s, nb(ni_s), dim3(256), 0, st, ni_s, T, key_s2, rs_s, d_ss, r_sp, r_d0, vrow, S_other, t_d0, S_side);
    hipLaunchKernelGGL(k_dp_fill_d, nb(ni_d), dim3(256), 0, st, ni_d, T, key_d2, rs_d, d_sd, r_dm, r_w, vrow, D_o, t_w, D_role);
    hipLaunchKernelGGL((k_dp_halo<0>), dim3(n_tiles), dim3(256), 0, st, d.tile_rows, T, (uint16_t*)nullptr, d_ss, d_sd, S_other, S_side, D_o, D_role, hs, hns, (const int*)nullptr,
                       (int*)nullptr, (uint32_t*)nullptr, (uint2*)nullptr, d_flag);
    std::vector<int> h_hs(n_tiles), h_hns(n_tiles), h_vrow(M);
    int h_flag = 0;
    NRS_HIP(c, hipMemcpyAsync(h_hs.data(), hs, sizeof(int) * n_tiles, hipMemcpyDeviceToHost, st));
    NRS_HIP(c, hipMemcpyAsync(h_hns.data(), hns, sizeof(int) * n_tiles, hipMemcpyDeviceToHost, st));
    NRS_HIP(c, hipMemcpyAsync(&h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
    NRS_HIP(c, hipMemcpyAsync(h_vrow.data(), vrow, sizeof(int) * (size_t)M, hipMemcpyDeviceToHost, st));
    NRS_HIP(c, hipStreamSynchronize(st));
    mark("fill + halo sizes");
    if (h_flag) return NRS_OK;                                       // a halo beyond the kernel's limits: host path
    // ---- halo_ptr, tile classes, LDS decision (host, O(tiles): as engine_create)
    std::vector<int> halo_ptr(n_tiles + 1, 0), tile_list(n_tiles);
    d.max_halo = 0; d.max_halo_s = 0;
    for (int b = 0; b < n_tiles; ++b) {
        halo_ptr[b + 1] = halo_ptr[b] + h_hs[b];
        d.max_halo = std::max(d.max_halo, h_hs[b]);
        d.max_halo_s = std::max(d.max_halo_s, h_hns[b]);
    }
    const size_t n_halo = (size_t)halo_ptr[n_tiles];
    {
        std::vector<int> sorted = h_hs;
        std::sort(sorted.begin(), sorted.end());
        int cut = d.max_halo;
        if (n_tiles >= 1024) {                                      // small problems are latency-bound: one launch
            const int p97 = sorted[(size_t)(0.97 * (n_tiles - 1))];
            const bool fits = sizeof(double) * 3 * (size_t)(2 * d.tile_rows + d.max_halo + d.max_halo_s + 2) <= 48 * 1024;
            if (4 * d.max_halo > 5 * p97 && (n_tiles - (int)(0.97 * n_tiles) >= 1024 || !fits) && !c->env("NRS_ONE_CLASS")) cut = p97;
        }
        int n0 = 0;
        for (int b = 0; b < n_tiles; ++b) if (h_hs[b] <= cut) tile_list[n0++] = b;
        int n1 = n0;
        for (int b = 0; b < n_tiles; ++b) if (h_hs[b] > cut) tile_list[n1++] = b;
        d.n_tiles_cls[0] = n0; d.n_tiles_cls[1] = n_tiles - n0;
        for (int b = 0; b < n_tiles; ++b) {
            const int cls = h_hs[b] <= cut ? 0 : 1;
            d.cap_h[cls] = std::max(d.cap_h[cls], h_hs[b]);
            d.cap_s[cls] = std::max(d.cap_s[cls], h_hns[b]);
        }
    }
    size_t lds_need = 0;
    for (int cls = 0; cls < 2; ++cls) {
        if (!d.n_tiles_cls[cls]) continue;
        lds_need = std::max(lds_need, sizeof(double) * 3 * (size_t)(d.tile_rows + d.cap_h[cls]));
        lds_need = std::max(lds_need, sizeof(double) * 3 * (size_t)(2 * d.tile_rows + d.cap_h[cls] + d.cap_s[cls] + 2));
    }
    if (lds_need > 64 * 1024 - 512 || d.tile_rows + d.max_halo >= 65535) return NRS_OK;     // gather fallback: host path
    d.use_lds = 1; d.dform = 0; d.coarse = 0;
    d.fused = d.n_rows < 32768 ? 1 : 0;                             // single-launch PCG iteration for latency-bound windows
    d.hier = n_tiles > 4096 ? 1 : 0;
    d.ecd = (!d.fused && !c->opt.profile) ? 1 : 0;
    d.co_n = 3 * d.n_groups + 6;
    d.sh_on = 0; d.sh_rank = 0; d.sh_world = 1; d.sh_lead = 1;
    d.sh_k0 = 0; d.sh_nk = K; d.sh_g0 = 0; d.sh_ng = d.n_groups; d.sh_vb0 = 0; d.sh_nvb = d.n_vecblk;
    for (int cls = 0; cls < 2; ++cls) { d.sh_t0[cls] = 0; d.sh_nt[cls] = d.n_tiles_cls[cls]; }
    d.ec_on = 1; d.plain = 1;
    d.lin_rb = ROW_ALIGN / (64 / T);
    d.ec_nsp = n_sp; d.ec_ndm = n_dm;
    d.ec_nblk = std::min((n_sp + n_dm + BLK - 1) / BLK, 2048);
    e->pack_rows = n_rows;
    if (tm) fprintf(stderr, "[nrs] device pack: tiles %d x %d rows, halo rows: max %d, mean %.1f, spring part max %d, classes %d (cap %d/%d) + %d (cap %d/%d)\n", n_tiles, d.tile_rows,
                    d.max_halo, (double)n_halo / n_tiles, d.max_halo_s, d.n_tiles_cls[0], d.cap_h[0], d.cap_s[0], d.n_tiles_cls[1], d.cap_h[1], d.cap_s[1]);
    // ---- arena
    ArenaPlan dry{arena, true};
    {
        Dev tmpd = d;
        Engine te;
        carve(dry, tmpd, false, nnz_s, nnz_d, (size_t)n_slices, n_halo, &te);
    }
    if (dry.off > arena->cap) {
        arena_release(arena);
        const size_t want = dry.off + dry.off / 8;
        hipError_t he = hipMalloc((void**)&arena->base, want);
        if (he != hipSuccess) return c->fail(NRS_ERR_ALLOC, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(he));
        arena->cap = want;
        if (c->env("NRS_POISON")) { (void)hipMemset(arena->base, 0xFF, want); (void)hipDeviceSynchronize(); }    // (debug: a read of memory nobody wrote shows up as NaN)
    }
    ArenaPlan real{arena, false};
    carve(real, d, false, nnz_s, nnz_d, (size_t)n_slices, n_halo, e);
    e->arena_bytes = real.off;
    mark("arena");
    // ---- final arrays
    NRS_HIP(c, hipMemcpyAsync(d.grp_pose, grp_pose.data(), sizeof(int) * grp_pose.size(), hipMemcpyHostToDevice, st));
    NRS_HIP(c, hipMemcpyAsync(d.pose_grp_ptr, pose_grp_ptr.data(), sizeof(int) * (K + 1), hipMemcpyHostToDevice, st));
    NRS_HIP(c, hipMemcpyAsync(d.halo_ptr, halo_ptr.data(), sizeof(int) * (n_tiles + 1), hipMemcpyHostToDevice, st));
    NRS_HIP(c, hipMemcpyAsync(d.tile_list, tile_list.data(), sizeof(int) * n_tiles, hipMemcpyHostToDevice, st));
    NRS_HIP(c, hipMemcpyAsync(d.halo_ns, hns, sizeof(int) * n_tiles, hipMemcpyDeviceToDevice, st));
    NRS_HIP(c, hipMemcpyAsync(d.ss_ptr, d_ss, sizeof(int) * ((size_t)n_slices + 1), hipMemcpyDeviceToDevice, st));
    NRS_HIP(c, hipMemcpyAsync(d.sd_ptr, d_sd, sizeof(int) * ((size_t)n_slices + 1), hipMemcpyDeviceToDevice, st));
    NRS_HIP(c, hipMemcpyAsync(d.s_d0, t_d0, sizeof(float) * nnz_s, hipMemcpyDeviceToDevice, st));
    NRS_HIP(c, hipMemcpyAsync(d.d_w, t_w, sizeof(float) * nnz_d, hipMemcpyDeviceToDevice, st));
    std::vector<Pose> poses(s.poses, s.poses + K);
    NRS_HIP(c, hipMemcpyAsync(d.pose_init, poses.data(), sizeof(Pose) * K, hipMemcpyHostToDevice, st));
    NRS_HIP(c, hipMemsetAsync(d.pose_fixed, 0, K, st));
    NRS_HIP(c, hipMemsetAsync(d.row_tp, 0xFF, sizeof(uint32_t) * (size_t)n_rows, st));
    hipLaunchKernelGGL((k_dp_halo<1>), dim3(n_tiles), dim3(256), 0, st, d.tile_rows, T, reinterpret_cast<uint16_t*>(d.row_tp), d.ss_ptr, d.sd_ptr, S_other, S_side, D_o, D_role, hs, hns, d.halo_ptr,
                       d.halo_rows, d.s_om, d.d_hdr, d_flag);
    if (d.fused) {
        std::vector<int> tile_desc(8 * (size_t)n_tiles, 0);
        const int rb = ROW_ALIGN / d.tile_rows;
        for (int b = 0; b < n_tiles; ++b) {
            const int kf = grp_pose[(size_t)b * d.tile_rows / ROW_ALIGN];
            int* td = &tile_desc[8 * (size_t)b];
            td[0] = kf; td[1] = pose_grp_ptr[kf] * rb; td[2] = pose_grp_ptr[kf + 1] * rb;
            td[3] = halo_ptr[b]; td[4] = halo_ptr[b + 1] - halo_ptr[b];
        }
        NRS_HIP(c, hipMemcpyAsync(d.tile_desc, tile_desc.data(), sizeof(int) * tile_desc.size(), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_dp_halofix, dim3(n_tiles), dim3(BLK), 0, st, n_tiles, d.halo_ptr, d.halo_rows, d.halo_fix, BLK, 0);
        NRS_HIP(c, hipStreamSynchronize(st));                        // (tile_desc dies here)
    } else if (d.use_lds)
        hipLaunchKernelGGL(k_dp_halofix, dim3(n_tiles), dim3(BLK), 0, st, n_tiles, d.halo_ptr, d.halo_rows, d.halo_fix, HALO_FIX, -1);
    hipLaunchKernelGGL(k_dp_rowdata, nb(n_rows), dim3(256), 0, st, n_rows, row_v, r_x, r_uv, d.rflag, d.uv, d.xl_init);
    hipLaunchKernelGGL(k_dp_rowcnt, nb(n_rows), dim3(256), 0, st, n_rows, cnt_s, cnt_d, d.row_cnt);
    hipLaunchKernelGGL(k_dp_eckeys, nb(std::max(n_sp, n_dm)), dim3(256), 0, st, n_sp, r_sp, n_dm, r_dm, vrow, ek_s, ek_d);
    tb = tmp_bytes; NRS_HIP(c, rocprim::radix_sort_keys(tmp, tb, ek_s, ek_s2, (size_t)n_sp, 0, 32 + row_bits, st));
    tb = tmp_bytes; NRS_HIP(c, rocprim::radix_sort_keys(tmp, tb, ek_d, ek_d2, (size_t)n_dm, 0, 32 + row_bits, st));
    hipLaunchKernelGGL(k_dp_ecfill, nb(std::max(n_sp, n_dm)), dim3(256), 0, st, n_sp, ek_s2, r_sp, r_d0, n_dm, ek_d2, r_dm, r_w, vrow, d.ec_sp, d.ec_dm, d.ec_w);
    NRS_HIP(c, hipMemsetAsync(d.s_qc, 0, sizeof(double) * nnz_s, st));
    NRS_HIP(c, hipMemsetAsync(d.d_s, 0, sizeof(double) * nnz_d, st));
    NRS_HIP(c, hipMemsetAsync(d.part_apply, 0, sizeof(double) * (size_t)d.n_vecblk, st));
    NRS_HIP(c, hipMemsetAsync(d.scal, 0, sizeof(double) * SC_N, st));
    NRS_HIP(c, hipMemsetAsync(d.flags, 0, sizeof(int) * 8, st));
    NRS_HIP(c, hipGetLastError());
    // ---- host side of the engine
    e->vrow.swap(h_vrow);
    e->h_rflag.assign(n_rows, RF_FIXED);
    for (int v = 0; v < M; ++v) e->h_rflag[e->vrow[v]] = RF_OBS | RF_REPROJ_ACTIVE;
    e->h_pose_fixed.assign(K, 0);
    e->dev_edges = true;                                             // residual taps copy the raw edges from the pack scratch (device to device)
    e->raw_sp = r_sp; e->raw_d0 = r_d0; e->raw_dm = r_dm; e->raw_w = r_w;
    e->serial = ++c->engine_serial;
    if (!c->pin_scal) NRS_HIP(c, hipHostMalloc((void**)&c->pin_scal, sizeof(double) * SC_N, hipHostMallocMapped | hipHostMallocCoherent));
    if (!c->pin_flags) {
        NRS_HIP(c, hipHostMalloc((void**)&c->pin_flags, sizeof(int) * 8, hipHostMallocMapped | hipHostMallocCoherent));
        memset(c->pin_flags, 0, sizeof(int) * 8);
    }
    e->h_scal = c->pin_scal; e->h_flags = c->pin_flags;
    d.h_scal = c->pin_scal; d.h_flags = c->pin_flags;
    NRS_HIP(c, hipStreamSynchronize(st));                            // (host staging vectors die here; the overflow word is final)
    int h_fl2[2] = {0, 0};
    NRS_HIP(c, hipMemcpy(h_fl2, d_flag, sizeof(int) * 2, hipMemcpyDeviceToHost));
    if (h_fl2[0]) return c->fail(NRS_ERR_HIP, "device pack: halo hash overflow in the second pass");
    d.tp_ok = h_fl2[1] ? 0 : 1;
    if (c->pack_ws2.cap > ((size_t)512 << 20)) c->release(c->pack_ws2);      // intermediates of a large window: not worth keeping resident
    mark("final arrays");
    engine_compact_headers(c, e);
    NRS_TRY(engine_reset(c, e));
    *done = true;
    return NRS_OK;
}

}  // namespace nrs
