// a3: LocalDeformableBundleAdjustment on MI355X
// (reference modules/optimization/g2o_optimization.cc:880-1161 + the g2o machinery it drives:
//  third_party/g2o/g2o/core/{optimization_algorithm_levenberg.cpp:57-174, block_solver.hpp:495-562,
//  base_fixed_sized_edge.hpp:49-133, robust_kernel_impl.cpp:60-74}).
//
// Design (MI355X-first, not a translation of g2o's pointer graph):
//   * unknowns: K pose blocks (6) then one 3-dof block per landmark ROW.  Landmarks are laid out
//     keyframe-major and every keyframe is padded to a multiple of ROW_ALIGN rows, so a workgroup
//     never straddles two keyframes and the per-pose reductions are plain block reductions.
//   * every edge is owned by the rows it touches ("incidence lists", sliced-ELL layout: a wave64
//     serves 64/T rows with T lanes per row; element (row r, lane t, step j) of a slice lives at
//     base + j*64 + r*T + t so all loads of a wave are one contiguous 64-element run).  Nothing
//     is scattered: no atomics, no assembly maps, bit-reproducible sums.
//   * H is never assembled.  The damped normal equations (H + lambda I) x = b are solved by a
//     block-Jacobi preconditioned conjugate gradient (single-reduction Chronopoulos-Gear form:
//     2 launches per iteration) whose operator is applied from per-incidence factors:
//         spring  : q g g^T is rank one          -> store g~ = sqrt(q) g      (24 B)
//         damper  : all 16 blocks are +-s I3      -> store s                   (8 B)
//         reproj  : H_pl (6x3), H_ll in the row's 3x3 diagonal block, H_pp reduced per keyframe
//     The reference factorises the same matrix with a sparse Cholesky (no Schur: H_ll is not
//     block diagonal); PCG to 1e-10 relative residual reproduces its iterates.
//   * LM control flow (lambda schedule, accept/reject, <=10 trials) runs on the host exactly as in
//     g2o; per trial the host reads back two scalars.  PCG convergence is detected on device; the
//     host only polls a flag once per batch of iterations.
//   * XCD-aware launch order: logical tile = (blockIdx % 8) * ceil(nb/8) + blockIdx / 8, so each
//     XCD's L2 serves one contiguous run of rows (neighbour gathers stay inside it).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <new>
#include "nrs_ctx.hpp"
#include "nrs_device.hpp"

namespace nrs {

constexpr int ROW_ALIGN = 256;       // keyframe row padding; also rows per k_reproj workgroup
constexpr int BLK = 256;             // threads per workgroup everywhere
constexpr int NPART = 12;            // per-block partial slots of the SpMV kernel

struct Sell {                        // sliced-ELL incidence storage (device pointers)
    int* slice_ptr;                  // n_slices+1, element offsets (multiples of 64)
    int nnz;                         // total padded elements
};

struct DbaDev {
    int K, M, n_rows, n_groups;      // poses, landmarks, padded rows, ROW_ALIGN groups
    int T;                           // lanes per row
    int n_sp, n_dm;
    Cam cam;
    double info_reproj, delta_reproj, info_pos, delta_pos, info_spatial, delta_spatial, k_spring;
    int spring_form;                 // 0 = BA Jacobian as written (position_regularizer.cc:51-60)
    // rows
    int* grp_kf;                     // n_groups -> pose index
    int* kf_grp_ptr;                 // K+1 -> group ranges
    int* row_lm;                     // n_rows -> landmark index or -1 (padding)
    float* uv;                       // n_rows x 2
    // incidences
    Sell ss, sd;
    int* s_other; float* s_d0; int* s_meta;
    int* d_o0; int* d_o1; int* d_o2; float* d_w; int* d_meta;
    // state (two copies: current / trial, swapped on accept)
    Pose* pose[2]; double* xl[2];
    Pose* pose_init; double* xl_init;
    // linearisation
    double* D;                       // n_rows x 6   (xx xy xz yy yz zz)
    double* Hpl;                     // 18 x n_rows  (component-major)
    double* s_g;                     // 3 x nnz_s
    double* d_s;                     // nnz_d
    double* Hpp;                     // K x 21
    double* bp; double* bl;          // 6K, 3 n_rows
    double* Dinv; double* Hppinv;    // n_rows x 6, K x 36
    // PCG vectors: pose part [6K] and row part [3 n_rows]
    double *xp, *rp, *up, *pp, *sp, *wp;
    double *xv, *rv, *uv3, *pv, *sv, *wv;
    // partials / scalars
    double* part_lin;                // n_groups x 32   (reproj kernel: 27 pose sums + chi)
    double* part_reg;                // n_regblk x 2    (chi, maxdiag)
    double* part_spmv;               // n_regblk x NPART
    double* part_apply;              // n_regblk x 1
    double* scal;                    // misc device scalars (see enum)
    int* flags;                      // [0] pcg done, [1] pcg iterations, [2] nan flag
    int n_regblk;                    // workgroups of the T-lane kernels
    int n_vecblk;                    // workgroups of the one-thread-per-row kernels
};

enum { SC_CHI = 0, SC_MAXDIAG = 1, SC_SCALE = 2, SC_GAMMA0 = 3, SC_SLOT0 = 4 /* gamma_old, alpha_old */, SC_SLOT1 = 6, SC_N = 16 };

struct DbaProblem {
    DbaDev d;
    std::vector<void*> allocs;
    std::vector<int> kf_ptr;         // K+1 landmark ranges
    std::vector<int> lm_row;         // landmark -> row
    int cur = 0;                     // which state copy is current
    double* h_scal = nullptr;        // pinned host mirror
    int* h_flags = nullptr;
    std::vector<int> sp_ij, dm_idx;  // host copies for the residual tap
    std::vector<float> sp_d0, dm_w;
};

// =====================================================================================
// device helpers
// =====================================================================================
__device__ inline int xcd_tile(int b, int nb) {
    const int nb8 = (nb + 7) >> 3;
    return (b & 7) * nb8 + (b >> 3);
}

template <int N>
__device__ inline void block_sum(double* v, double* lds /* 4*N */, int lane, int wave) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double s = wave_sum(v[k]);
        if (lane == 0) lds[wave * N + k] = s;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = lds[k] + lds[N + k] + lds[2 * N + k] + lds[3 * N + k];
    __syncthreads();
}

// same reduction, totals written to out[0..N) by the first N threads (avoids dynamic register indexing)
template <int N>
__device__ inline void block_sum_store(const double* v, double* lds /* 4*N */, int tid, double* out) {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double s = wave_sum(v[k]);
        if (lane == 0) lds[wave * N + k] = s;
    }
    __syncthreads();
    if (tid < N) out[tid] = lds[tid] + lds[N + tid] + lds[2 * N + tid] + lds[3 * N + tid];
}

__device__ inline double sub_sum(double v, int T) {      // reduce over the T lanes of a row
    for (int off = 1; off < T; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ inline bool inv3_sym(const double* d /*xx xy xz yy yz zz*/, double lam, double* o) {
    const double a = d[0] + lam, b = d[1], c = d[2], e = d[3] + lam, f = d[4], g = d[5] + lam;
    const double c00 = e * g - f * f, c01 = c * f - b * g, c02 = b * f - c * e;
    const double det = a * c00 + b * c01 + c * c02;
    const double id = 1.0 / det;
    o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
    o[3] = (a * g - c * c) * id; o[4] = (b * c - a * f) * id; o[5] = (a * e - b * b) * id;
    return det > 0;
}

// =====================================================================================
// linearisation, part 1: reprojection edges.  One thread per row, one keyframe per workgroup.
//   ReprojectionError::computeError / linearizeOplus (reference reprojection_error.cc:32-64),
//   quadratic form with Huber weight (base_fixed_sized_edge.hpp:49-63, base_edge.h:158-164).
// =====================================================================================
template <bool LIN>
__global__ __launch_bounds__(BLK) void k_reproj(DbaDev P, const Pose* __restrict__ poses,
                                                const double* __restrict__ xl) {
    __shared__ double lds[4 * 28];
    const int g = xcd_tile(blockIdx.x, P.n_groups);
    if (g >= P.n_groups) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = g * ROW_ALIGN + tid;
    const int kf = P.grp_kf[g];
    const Pose T = poses[kf];
    double R[9];
    quat_to_R(T.q, R);
    double acc[28];
#pragma unroll
    for (int k = 0; k < 28; ++k) acc[k] = 0;
    const bool valid = P.row_lm[row] >= 0;
    if (valid) {
        const double x0 = xl[3 * row], x1 = xl[3 * row + 1], x2 = xl[3 * row + 2];
        const double px = R[0] * x0 + R[1] * x1 + R[2] * x2 + T.t[0];
        const double py = R[3] * x0 + R[4] * x1 + R[5] * x2 + T.t[1];
        const double pz = R[6] * x0 + R[7] * x1 + R[8] * x2 + T.t[2];
        float u, v;
        project_f32(P.cam, (float)px, (float)py, (float)pz, u, v);
        const double r0 = (double)P.uv[2 * row] - (double)u, r1 = (double)P.uv[2 * row + 1] - (double)v;
        double rho0, rho1;
        huber(P.info_reproj * (r0 * r0 + r1 * r1), P.delta_reproj, rho0, rho1);
        acc[27] = rho0;
        if (LIN) {
            float Jf[6];
            projection_jacobian_f32(P.cam, (float)px, (float)py, (float)pz, Jf);
            const double w = rho1 * P.info_reproj;
            double Jp[2][6], Jl[2][3];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const double j0 = -(double)Jf[3 * rr], j1 = -(double)Jf[3 * rr + 1], j2 = -(double)Jf[3 * rr + 2];
                Jp[rr][0] = -j1 * pz + j2 * py;
                Jp[rr][1] = j0 * pz - j2 * px;
                Jp[rr][2] = -j0 * py + j1 * px;
                Jp[rr][3] = j0; Jp[rr][4] = j1; Jp[rr][5] = j2;
                Jl[rr][0] = j0 * R[0] + j1 * R[3] + j2 * R[6];
                Jl[rr][1] = j0 * R[1] + j1 * R[4] + j2 * R[7];
                Jl[rr][2] = j0 * R[2] + j1 * R[5] + j2 * R[8];
            }
            int k = 0;
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int q = p; q < 6; ++q) { acc[k] = w * (Jp[0][p] * Jp[0][q] + Jp[1][p] * Jp[1][q]); ++k; }
#pragma unroll
            for (int p = 0; p < 6; ++p) acc[21 + p] = -w * (Jp[0][p] * r0 + Jp[1][p] * r1);
            // H_pl (6x3), component-major so that a wave writes 18 contiguous runs
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    P.Hpl[(size_t)(p * 3 + c) * P.n_rows + row] = w * (Jp[0][p] * Jl[0][c] + Jp[1][p] * Jl[1][c]);
            double* D = P.D + 6 * (size_t)row;
            D[0] = w * (Jl[0][0] * Jl[0][0] + Jl[1][0] * Jl[1][0]);
            D[1] = w * (Jl[0][0] * Jl[0][1] + Jl[1][0] * Jl[1][1]);
            D[2] = w * (Jl[0][0] * Jl[0][2] + Jl[1][0] * Jl[1][2]);
            D[3] = w * (Jl[0][1] * Jl[0][1] + Jl[1][1] * Jl[1][1]);
            D[4] = w * (Jl[0][1] * Jl[0][2] + Jl[1][1] * Jl[1][2]);
            D[5] = w * (Jl[0][2] * Jl[0][2] + Jl[1][2] * Jl[1][2]);
            P.bl[3 * row] = -w * (Jl[0][0] * r0 + Jl[1][0] * r1);
            P.bl[3 * row + 1] = -w * (Jl[0][1] * r0 + Jl[1][1] * r1);
            P.bl[3 * row + 2] = -w * (Jl[0][2] * r0 + Jl[1][2] * r1);
        }
    } else if (LIN) {
#pragma unroll
        for (int c = 0; c < 18; ++c) P.Hpl[(size_t)c * P.n_rows + row] = 0;
#pragma unroll
        for (int c = 0; c < 6; ++c) P.D[6 * (size_t)row + c] = 0;
        P.bl[3 * row] = P.bl[3 * row + 1] = P.bl[3 * row + 2] = 0;
    }
    if (LIN) {
        block_sum_store<28>(acc, lds, tid, P.part_lin + (size_t)g * 32);
    } else {
        double c1[1] = {acc[27]};
        block_sum<1>(c1, lds, lane, wave);
        if (tid == 0) P.part_lin[(size_t)g * 32 + 27] = c1[0];
    }
}

// =====================================================================================
// linearisation, part 2: springs and dampers from the incidence lists (T lanes per row).
//   PositionRegularizer (position_regularizer.cc:32-61, Jacobian as written),
//   SpatialRegularizer  (spatial_regularizer.cc:32-59).
// =====================================================================================
template <int T, bool LIN>
__global__ __launch_bounds__(BLK) void k_reg(DbaDev P, const double* __restrict__ xl) {
    __shared__ double lds[4 * 2];
    constexpr int R = 64 / T;
    const int b = xcd_tile(blockIdx.x, P.n_regblk);
    if (b >= P.n_regblk) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slice = b * 4 + wave;
    const int row = slice * R + lane / T;
    const int t = lane % T;
    const double xo0 = xl[3 * row], xo1 = xl[3 * row + 1], xo2 = xl[3 * row + 2];
    double D[6] = {0, 0, 0, 0, 0, 0}, bb[3] = {0, 0, 0}, chi = 0;
    // ---- springs
    {
        const int beg = P.ss.slice_ptr[slice], end = P.ss.slice_ptr[slice + 1];
        for (int idx = beg + lane; idx < end; idx += 64) {
            const int o = P.s_other[idx];
            if (o < 0) continue;
            const double d0 = (double)P.s_d0[idx];
            const double v0 = xo0 - xl[3 * o], v1 = xo1 - xl[3 * o + 1], v2 = xo2 - xl[3 * o + 2];
            const double d = sqrt(v0 * v0 + v1 * v1 + v2 * v2);
            const double r = P.k_spring * (d - d0) / d0;
            double rho0, rho1;
            huber(P.info_pos * r * r, P.delta_pos, rho0, rho1);
            if (P.s_meta[idx] >> 30) chi += rho0;
            if (LIN) {
                const double cg = P.spring_form == 0 ? (P.k_spring / d0) * (1.0 / sqrt(d)) * 2.0
                                                     : P.k_spring / (d0 * d);
                const double q = rho1 * P.info_pos;
                const double g0 = cg * v0, g1 = cg * v1, g2 = cg * v2;
                const double sq = sqrt(q);
                P.s_g[idx] = sq * g0;
                P.s_g[(size_t)P.ss.nnz + idx] = sq * g1;
                P.s_g[2 * (size_t)P.ss.nnz + idx] = sq * g2;
                D[0] += q * g0 * g0; D[1] += q * g0 * g1; D[2] += q * g0 * g2;
                D[3] += q * g1 * g1; D[4] += q * g1 * g2; D[5] += q * g2 * g2;
                const double qr = q * r;
                bb[0] -= qr * g0; bb[1] -= qr * g1; bb[2] -= qr * g2;
            }
        }
    }
    // ---- dampers: r = w((x1n - x1c) - (x2n - x2c)), roles (1c,2c,1n,2n), signs (-,+,+,-)
    {
        const int beg = P.sd.slice_ptr[slice], end = P.sd.slice_ptr[slice + 1];
        for (int idx = beg + lane; idx < end; idx += 64) {
            const int meta = P.d_meta[idx];
            if (meta < 0) continue;
            const int role = meta & 3;
            const int o[3] = {P.d_o0[idx], P.d_o1[idx], P.d_o2[idx]};
            const double w = (double)P.d_w[idx];
            const double sgn_own = (role == 0 || role == 3) ? -1.0 : 1.0;
            double s0 = sgn_own * xo0, s1 = sgn_own * xo1, s2 = sgn_own * xo2;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int ro = k + (k >= role ? 1 : 0);          // role of the k-th other vertex
                const double sg = (ro == 0 || ro == 3) ? -1.0 : 1.0;
                if (o[k] >= 0) {
                    s0 += sg * xl[3 * o[k]]; s1 += sg * xl[3 * o[k] + 1]; s2 += sg * xl[3 * o[k] + 2];
                }
            }
            const double r0 = w * s0, r1 = w * s1, r2 = w * s2;
            double rho0, rho1;
            huber(P.info_spatial * (r0 * r0 + r1 * r1 + r2 * r2), P.delta_spatial, rho0, rho1);
            if ((meta >> 2) & 1) chi += rho0;
            if (LIN) {
                const double s = rho1 * P.info_spatial * w * w;
                P.d_s[idx] = s;
                D[0] += s; D[3] += s; D[5] += s;
                const double c = sgn_own * rho1 * P.info_spatial * w;
                bb[0] -= c * r0; bb[1] -= c * r1; bb[2] -= c * r2;
            }
        }
    }
    double part[2];
    part[0] = chi;
    part[1] = 0;
    if (LIN) {
#pragma unroll
        for (int k = 0; k < 6; ++k) D[k] = sub_sum(D[k], T);
#pragma unroll
        for (int k = 0; k < 3; ++k) bb[k] = sub_sum(bb[k], T);
        if (t == 0) {
            double* Dr = P.D + 6 * (size_t)row;
            double dd[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) { dd[k] = Dr[k] + D[k]; Dr[k] = dd[k]; }
#pragma unroll
            for (int k = 0; k < 3; ++k) P.bl[3 * row + k] += bb[k];
            part[1] = fmax(fabs(dd[0]), fmax(fabs(dd[3]), fabs(dd[5])));
        }
    }
    // chi: sum; maxdiag: max  (max via wave butterflies)
    double c = wave_sum(part[0]);
    double m = part[1];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
    if (lane == 0) { lds[wave * 2] = c; lds[wave * 2 + 1] = m; }
    __syncthreads();
    if (tid == 0) {
        P.part_reg[2 * (size_t)b] = lds[0] + lds[2] + lds[4] + lds[6];
        P.part_reg[2 * (size_t)b + 1] = fmax(fmax(lds[1], lds[3]), fmax(lds[5], lds[7]));
    }
}

// =====================================================================================
// finalize: fixed-order sums of the partials.  LIN: H_pp, b_p, chi2, max diag.  else chi2 only.
// =====================================================================================
template <bool LIN>
__global__ __launch_bounds__(BLK) void k_finalize(DbaDev P) {
    __shared__ double lds[4 * 3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double chi = 0, md = 0, sc = 0;
    if (!LIN)
        for (int b = tid; b < P.n_vecblk; b += BLK) sc += P.part_apply[b];
    for (int g = tid; g < P.n_groups; g += BLK) chi += P.part_lin[(size_t)g * 32 + 27];
    for (int b = tid; b < P.n_regblk; b += BLK) {
        chi += P.part_reg[2 * (size_t)b];
        md = fmax(md, P.part_reg[2 * (size_t)b + 1]);
    }
    if (LIN) {
        for (int i = tid; i < P.K * 27; i += BLK) {
            const int k = i / 27, c = i % 27;
            double s = 0;
            for (int g = P.kf_grp_ptr[k]; g < P.kf_grp_ptr[k + 1]; ++g) s += P.part_lin[(size_t)g * 32 + c];
            if (c < 21) {
                P.Hpp[k * 21 + c] = s;
                // diagonal entries of the packed upper triangle: 0,6,11,15,18,20
                if (c == 0 || c == 6 || c == 11 || c == 15 || c == 18 || c == 20) md = fmax(md, fabs(s));
            } else {
                P.bp[k * 6 + (c - 21)] = s;
            }
        }
    }
    double c = wave_sum(chi);
    sc = wave_sum(sc);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) md = fmax(md, __shfl_xor(md, off, 64));
    if (lane == 0) { lds[wave * 3] = c; lds[wave * 3 + 1] = md; lds[wave * 3 + 2] = sc; }
    __syncthreads();
    if (tid == 0) {
        P.scal[SC_CHI] = lds[0] + lds[3] + lds[6] + lds[9];
        if (LIN) P.scal[SC_MAXDIAG] = fmax(fmax(lds[1], lds[4]), fmax(lds[7], lds[10]));
        if (!LIN) P.scal[SC_SCALE] = lds[2] + lds[5] + lds[8] + lds[11];
    }
}

// =====================================================================================
// per-trial setup: block-Jacobi preconditioner for (H + lambda I) and the PCG start vectors
//   x = 0, r = b, u = M^-1 r, p = s = 0.
// =====================================================================================
__device__ inline bool inv6_spd(const double* Hu, double lam, double* Ainv /*36*/) {
    double L[6][6];
    double A[6][6];
    int k = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) { A[i][j] = Hu[k]; A[j][i] = Hu[k]; ++k; }
    for (int i = 0; i < 6; ++i) A[i][i] += lam;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) L[i][j] = 0;
    bool ok = true;
    for (int j = 0; j < 6; ++j) {
        double d = A[j][j];
        for (int q = 0; q < j; ++q) d -= L[j][q] * L[j][q];
        if (!(d > 0)) { ok = false; d = 1; }
        const double l = sqrt(d);
        L[j][j] = l;
        for (int i = j + 1; i < 6; ++i) {
            double s = A[i][j];
            for (int q = 0; q < j; ++q) s -= L[i][q] * L[j][q];
            L[i][j] = s / l;
        }
    }
    for (int c = 0; c < 6; ++c) {                       // solve A X = e_c
        double y[6], x[6];
        for (int i = 0; i < 6; ++i) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int q = 0; q < i; ++q) s -= L[i][q] * y[q];
            y[i] = s / L[i][i];
        }
        for (int i = 5; i >= 0; --i) {
            double s = y[i];
            for (int q = i + 1; q < 6; ++q) s -= L[q][i] * x[q];
            x[i] = s / L[i][i];
        }
        for (int i = 0; i < 6; ++i) Ainv[i * 6 + c] = x[i];
    }
    return ok;
}

__global__ __launch_bounds__(BLK) void k_trial_setup(DbaDev P, double lam) {
    const int i = blockIdx.x * BLK + threadIdx.x;
    if (i < P.n_rows) {
        double Di[6];
        const bool ok = inv3_sym(P.D + 6 * (size_t)i, lam, Di);
        if (!ok || !isfinite(Di[0])) P.flags[2] = 1;
#pragma unroll
        for (int k = 0; k < 6; ++k) P.Dinv[6 * (size_t)i + k] = Di[k];
        const double r0 = P.bl[3 * i], r1 = P.bl[3 * i + 1], r2 = P.bl[3 * i + 2];
        P.rv[3 * i] = r0; P.rv[3 * i + 1] = r1; P.rv[3 * i + 2] = r2;
        P.uv3[3 * i] = Di[0] * r0 + Di[1] * r1 + Di[2] * r2;
        P.uv3[3 * i + 1] = Di[1] * r0 + Di[3] * r1 + Di[4] * r2;
        P.uv3[3 * i + 2] = Di[2] * r0 + Di[4] * r1 + Di[5] * r2;
#pragma unroll
        for (int k = 0; k < 3; ++k) { P.xv[3 * i + k] = 0; P.pv[3 * i + k] = 0; P.sv[3 * i + k] = 0; }
    }
    if (i < P.K) {
        double Ai[36];
        if (!inv6_spd(P.Hpp + 21 * i, lam, Ai)) P.flags[2] = 1;
        for (int k = 0; k < 36; ++k) P.Hppinv[36 * i + k] = Ai[k];
        for (int a = 0; a < 6; ++a) {
            double s = 0;
            for (int c = 0; c < 6; ++c) s += Ai[a * 6 + c] * P.bp[6 * i + c];
            P.up[6 * i + a] = s;
            P.rp[6 * i + a] = P.bp[6 * i + a];
            P.xp[6 * i + a] = 0; P.pp[6 * i + a] = 0; P.sp[6 * i + a] = 0;
        }
    }
}

// =====================================================================================
// PCG kernel 1: w = (H + lambda I) u for the landmark rows, from the per-incidence factors,
// plus the per-block partials the update kernel needs:
//   [0] r.u  [1] w.u  [2] u_l.(H_pl^T u_p)  [3..8] H_pl u_l (pose rows)
// =====================================================================================
template <int T>
__global__ __launch_bounds__(BLK) void k_spmv(DbaDev P, double lam) {
    __shared__ double lds[4 * 9];
    if (P.flags[0]) return;
    constexpr int R = 64 / T;
    const int b = xcd_tile(blockIdx.x, P.n_regblk);
    if (b >= P.n_regblk) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slice = b * 4 + wave;
    const int row = slice * R + lane / T;
    const int t = lane % T;
    const double* __restrict__ u = P.uv3;
    double a0 = 0, a1 = 0, a2 = 0;
    double part[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double ul0 = 0, ul1 = 0, ul2 = 0;
    if (t == 0) {
        ul0 = u[3 * row]; ul1 = u[3 * row + 1]; ul2 = u[3 * row + 2];
        const int kf = P.grp_kf[row / ROW_ALIGN];
        const double* D = P.D + 6 * (size_t)row;
        a0 = (D[0] + lam) * ul0 + D[1] * ul1 + D[2] * ul2;
        a1 = D[1] * ul0 + (D[3] + lam) * ul1 + D[4] * ul2;
        a2 = D[2] * ul0 + D[4] * ul1 + (D[5] + lam) * ul2;
        double h0 = 0, h1 = 0, h2 = 0;
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            const double upk = P.up[6 * kf + p];
            const double e0 = P.Hpl[(size_t)(p * 3) * P.n_rows + row];
            const double e1 = P.Hpl[(size_t)(p * 3 + 1) * P.n_rows + row];
            const double e2 = P.Hpl[(size_t)(p * 3 + 2) * P.n_rows + row];
            h0 += e0 * upk; h1 += e1 * upk; h2 += e2 * upk;
            part[3 + p] = e0 * ul0 + e1 * ul1 + e2 * ul2;
        }
        a0 += h0; a1 += h1; a2 += h2;
        part[2] = ul0 * h0 + ul1 * h1 + ul2 * h2;
    }
    {
        const int beg = P.ss.slice_ptr[slice], end = P.ss.slice_ptr[slice + 1];
        const size_t nz = (size_t)P.ss.nnz;
        for (int idx = beg + lane; idx < end; idx += 64) {
            const int o = P.s_other[idx];
            if (o < 0) continue;
            const double g0 = P.s_g[idx], g1 = P.s_g[nz + idx], g2 = P.s_g[2 * nz + idx];
            const double dot = g0 * u[3 * o] + g1 * u[3 * o + 1] + g2 * u[3 * o + 2];
            a0 -= g0 * dot; a1 -= g1 * dot; a2 -= g2 * dot;
        }
    }
    {
        const int beg = P.sd.slice_ptr[slice], end = P.sd.slice_ptr[slice + 1];
        for (int idx = beg + lane; idx < end; idx += 64) {
            const int meta = P.d_meta[idx];
            if (meta < 0) continue;
            const int role = meta & 3;
            const int o[3] = {P.d_o0[idx], P.d_o1[idx], P.d_o2[idx]};
            double s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int ro = k + (k >= role ? 1 : 0);
                const double sg = (ro == 0 || ro == 3) ? -1.0 : 1.0;
                if (o[k] >= 0) { s0 += sg * u[3 * o[k]]; s1 += sg * u[3 * o[k] + 1]; s2 += sg * u[3 * o[k] + 2]; }
            }
            const double c = ((role == 0 || role == 3) ? -1.0 : 1.0) * P.d_s[idx];
            a0 += c * s0; a1 += c * s1; a2 += c * s2;
        }
    }
    a0 = sub_sum(a0, T); a1 = sub_sum(a1, T); a2 = sub_sum(a2, T);
    if (t == 0) {
        P.wv[3 * row] = a0; P.wv[3 * row + 1] = a1; P.wv[3 * row + 2] = a2;
        part[0] = P.rv[3 * row] * ul0 + P.rv[3 * row + 1] * ul1 + P.rv[3 * row + 2] * ul2;
        part[1] = a0 * ul0 + a1 * ul1 + a2 * ul2;
    }
    block_sum_store<9>(part, lds, tid, P.part_spmv + (size_t)b * NPART);
}

// =====================================================================================
// PCG kernel 2 (Chronopoulos-Gear single-reduction CG): every workgroup re-derives the scalars
// from the partials in a fixed order, then updates its rows:
//   gamma = r.u, delta = w.u, beta = gamma/gamma_old, alpha = gamma/(delta - beta*gamma/alpha_old)
//   p = u + beta p ; s = w + beta s ; x += alpha p ; r -= alpha s ; u = M^-1 r
// Workgroups >= n_vecblk own the pose rows (w_p = (H_pp + lambda) u_p + sum_l H_pl u_l).
// =====================================================================================
__global__ __launch_bounds__(BLK) void k_pcg_update(DbaDev P, double lam, int it, double tol2, int n_vecblk) {
    __shared__ double lds[4 * 3];
    if (P.flags[0]) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double v[3] = {0, 0, 0};
    for (int b = tid; b < P.n_regblk; b += BLK) {
        v[0] += P.part_spmv[(size_t)b * NPART];
        v[1] += P.part_spmv[(size_t)b * NPART + 1];
        v[2] += P.part_spmv[(size_t)b * NPART + 2];
    }
    // pose rows: gamma_p = r_p.u_p ; delta_p = u_p.(H_pp + lam)u_p + cross (cross is v[2])
    for (int i = tid; i < 6 * P.K; i += BLK) {
        const int k = i / 6, a = i % 6;
        const double ua = P.up[i];
        v[0] += P.rp[i] * ua;
        double s = lam * ua;
        for (int c = 0; c < 6; ++c) {
            const int lo = a < c ? a : c, hi = a < c ? c : a;
            const int pk = lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo);     // packed upper index
            s += P.Hpp[21 * k + pk] * P.up[6 * k + c];
        }
        v[1] += ua * s;
    }
    block_sum<3>(v, lds, lane, wave);
    const double gamma = v[0], delta = v[1] + v[2];
    const double* slot = P.scal + ((it & 1) ? SC_SLOT1 : SC_SLOT0);
    double* nslot = P.scal + ((it & 1) ? SC_SLOT0 : SC_SLOT1);
    const double gamma0 = it == 0 ? gamma : P.scal[SC_GAMMA0];
    const bool bad = !isfinite(gamma) || !isfinite(delta);
    const bool conv = (gamma <= tol2 * gamma0) || bad || gamma == 0.0;
    if (conv) {
        if (blockIdx.x == 0 && tid == 0) {
            if (bad) P.flags[2] = 1;
            P.flags[1] = it;
            __threadfence();
            P.flags[0] = 1;
        }
        return;
    }
    const double beta = it == 0 ? 0.0 : gamma / slot[0];
    const double alpha = it == 0 ? gamma / delta : gamma / (delta - beta * gamma / slot[1]);
    if (blockIdx.x == 0 && tid == 0) {
        nslot[0] = gamma;
        nslot[1] = alpha;
        if (it == 0) P.scal[SC_GAMMA0] = gamma;
        P.flags[1] = it + 1;
    }
    const int n_vec8 = ((n_vecblk + 7) >> 3) << 3;
    if ((int)blockIdx.x < n_vec8) {
        const int i = xcd_tile(blockIdx.x, n_vecblk) * BLK + tid;
        if (i < P.n_rows) {
            double r[3], uu[3];
            const double* Di = P.Dinv + 6 * (size_t)i;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int j = 3 * i + k;
                const double p = P.uv3[j] + beta * P.pv[j];
                const double s = P.wv[j] + beta * P.sv[j];
                P.pv[j] = p;
                P.sv[j] = s;
                P.xv[j] += alpha * p;
                r[k] = P.rv[j] - alpha * s;
                P.rv[j] = r[k];
            }
            uu[0] = Di[0] * r[0] + Di[1] * r[1] + Di[2] * r[2];
            uu[1] = Di[1] * r[0] + Di[3] * r[1] + Di[4] * r[2];
            uu[2] = Di[2] * r[0] + Di[4] * r[1] + Di[5] * r[2];
            P.uv3[3 * i] = uu[0]; P.uv3[3 * i + 1] = uu[1]; P.uv3[3 * i + 2] = uu[2];
        }
    } else {
        // pose workgroups: 42 poses per workgroup, 6 lanes per pose
        __shared__ double s_r[42 * 6];
        const int pb = blockIdx.x - n_vec8;
        const int kl = tid / 6, a = tid % 6;
        const int k = pb * 42 + kl;
        const bool act = kl < 42 && k < P.K;
        double rnew = 0;
        if (act) {
            const int i = 6 * k + a;
            const double ua = P.up[i];
            double w = lam * ua;
            for (int c = 0; c < 6; ++c) {
                const int lo = a < c ? a : c, hi = a < c ? c : a;
                const int pk = lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo);
                w += P.Hpp[21 * k + pk] * P.up[6 * k + c];
            }
            const int rb = ROW_ALIGN / (BLK / P.T);       // reg-blocks per row group
            for (int g = P.kf_grp_ptr[k] * rb; g < P.kf_grp_ptr[k + 1] * rb; ++g) w += P.part_spmv[(size_t)g * NPART + 3 + a];
            const double p = ua + beta * P.pp[i];
            const double s = w + beta * P.sp[i];
            P.pp[i] = p;
            P.sp[i] = s;
            P.xp[i] += alpha * p;
            rnew = P.rp[i] - alpha * s;
            P.rp[i] = rnew;
            s_r[kl * 6 + a] = rnew;
        }
        __syncthreads();
        if (act) {
            double s = 0;
            for (int c = 0; c < 6; ++c) s += P.Hppinv[36 * k + a * 6 + c] * s_r[kl * 6 + c];
            P.up[6 * k + a] = s;
        }
    }
}

// =====================================================================================
// trial state = state (+) x ;  partial of computeScale: sum_j x_j (lambda x_j + b_j)
// (levenberg.cpp:167-174; LandmarkVertex::oplusImpl landmark_vertex.cc:40-43)
// =====================================================================================
__global__ __launch_bounds__(BLK) void k_apply(DbaDev P, double lam, const Pose* __restrict__ pose_in,
                                               const double* __restrict__ xl_in, Pose* pose_out, double* xl_out) {
    __shared__ double lds[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x * BLK + tid;
    double sc[1] = {0};
    if (i < P.n_rows) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int j = 3 * i + k;
            const double x = P.xv[j];
            xl_out[j] = xl_in[j] + x;
            sc[0] += x * (lam * x + P.bl[j]);
        }
    }
    if (i < P.K) {
        Pose T = pose_in[i];
        double upd[6];
        for (int a = 0; a < 6; ++a) {
            upd[a] = P.xp[6 * i + a];
            sc[0] += upd[a] * (lam * upd[a] + P.bp[6 * i + a]);
        }
        pose_oplus(T, upd);
        pose_out[i] = T;
    }
    block_sum<1>(sc, lds, lane, wave);
    if (tid == 0) P.part_apply[blockIdx.x] = sc[0];
}

// =====================================================================================
// debug / parity taps (edge-parallel, not on the timed path)
// =====================================================================================
__global__ void k_tap_residuals(DbaDev P, const Pose* poses, const double* xl, const int* lm_row,
                                const int* sp_ij, const float* sp_d0, const int* dm_idx, const float* dm_w,
                                double* r_reproj, double* r_spring, double* r_damper) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P.M) {
        const int row = lm_row[i];
        const Pose T = poses[P.grp_kf[row / ROW_ALIGN]];
        double R[9];
        quat_to_R(T.q, R);
        const double x0 = xl[3 * row], x1 = xl[3 * row + 1], x2 = xl[3 * row + 2];
        const double px = R[0] * x0 + R[1] * x1 + R[2] * x2 + T.t[0];
        const double py = R[3] * x0 + R[4] * x1 + R[5] * x2 + T.t[1];
        const double pz = R[6] * x0 + R[7] * x1 + R[8] * x2 + T.t[2];
        float u, v;
        project_f32(P.cam, (float)px, (float)py, (float)pz, u, v);
        r_reproj[2 * i] = (double)P.uv[2 * row] - (double)u;
        r_reproj[2 * i + 1] = (double)P.uv[2 * row + 1] - (double)v;
    }
    if (i < P.n_sp) {
        const int a = lm_row[sp_ij[2 * i]], b = lm_row[sp_ij[2 * i + 1]];
        const double v0 = xl[3 * a] - xl[3 * b], v1 = xl[3 * a + 1] - xl[3 * b + 1], v2 = xl[3 * a + 2] - xl[3 * b + 2];
        const double d = sqrt(v0 * v0 + v1 * v1 + v2 * v2), d0 = (double)sp_d0[i];
        r_spring[i] = P.k_spring * (d - d0) / d0;
    }
    if (i < P.n_dm) {
        const int a = lm_row[dm_idx[4 * i]], b = lm_row[dm_idx[4 * i + 1]], c = lm_row[dm_idx[4 * i + 2]], d = lm_row[dm_idx[4 * i + 3]];
        const double w = (double)dm_w[i];
        for (int k = 0; k < 3; ++k)
            r_damper[3 * i + k] = w * ((xl[3 * c + k] - xl[3 * a + k]) - (xl[3 * d + k] - xl[3 * b + k]));
    }
}

// =====================================================================================
// host side
// =====================================================================================
static int dev_alloc(nrs_ctx* c, DbaProblem* pb, void** p, size_t bytes) {
    *p = nullptr;
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return c->fail(NRS_ERR_ALLOC, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    pb->allocs.push_back(*p);
    return NRS_OK;
}

template <class Tp>
static int dev_upload(nrs_ctx* c, DbaProblem* pb, Tp** dst, const std::vector<Tp>& src) {
    NRS_TRY(dev_alloc(c, pb, (void**)dst, sizeof(Tp) * src.size()));
    if (!src.empty()) NRS_HIP(c, hipMemcpyAsync(*dst, src.data(), sizeof(Tp) * src.size(), hipMemcpyHostToDevice, c->stream));
    return NRS_OK;
}

void dba_free(nrs_ctx* c) {
    if (!c->dba) return;
    (void)hipStreamSynchronize(c->stream);
    for (void* p : c->dba->allocs) (void)hipFree(p);
    if (c->dba->h_scal) (void)hipHostFree(c->dba->h_scal);
    if (c->dba->h_flags) (void)hipHostFree(c->dba->h_flags);
    delete c->dba;
    c->dba = nullptr;
}

struct Inc { int other[3]; float w; int meta; };

// pack per-row incidence lists into the sliced-ELL layout described at the top of the file
static void sell_pack(const std::vector<std::vector<Inc>>& rows, int T, std::vector<int>& slice_ptr,
                      std::vector<Inc>& out) {
    const int R = 64 / T;
    const int n_slices = (int)rows.size() / R;
    slice_ptr.assign(n_slices + 1, 0);
    for (int s = 0; s < n_slices; ++s) {
        int width = 0;
        for (int r = 0; r < R; ++r) width = std::max(width, ((int)rows[(size_t)s * R + r].size() + T - 1) / T);
        slice_ptr[s + 1] = slice_ptr[s] + width * 64;
    }
    Inc pad;
    pad.other[0] = pad.other[1] = pad.other[2] = -1;
    pad.w = 0;
    pad.meta = -1;
    out.assign((size_t)slice_ptr[n_slices], pad);
    for (int s = 0; s < n_slices; ++s)
        for (int r = 0; r < R; ++r) {
            const auto& L = rows[(size_t)s * R + r];
            for (size_t e = 0; e < L.size(); ++e) {
                const int j = (int)e / T, t = (int)e % T;
                out[(size_t)slice_ptr[s] + (size_t)j * 64 + r * T + t] = L[e];
            }
        }
}

static int dba_upload_impl(nrs_ctx* c, const nrs_camera* cam, int32_t n_kf, const double* poses_qt,
                           int32_t n_lm, const float* lm_xyz, const int32_t* lm_kf, const float* lm_uv,
                           int32_t n_sp, const int32_t* sp_ij, const float* sp_d0,
                           int32_t n_dm, const int32_t* dm_idx, const float* dm_w, float scale) {
    if (!cam || n_kf <= 0 || n_lm <= 0 || !poses_qt || !lm_xyz || !lm_kf || !lm_uv || n_sp < 0 || n_dm < 0 ||
        (n_sp > 0 && (!sp_ij || !sp_d0)) || (n_dm > 0 && (!dm_idx || !dm_w)))
        return c->fail(NRS_ERR_INVALID, "nrs_dba_upload: bad argument");
    if (cam->model != NRS_CAM_PINHOLE && cam->model != NRS_CAM_KB8) return c->fail(NRS_ERR_INVALID, "unknown camera model %d", cam->model);
    for (int i = 0; i < n_lm; ++i) {
        if (lm_kf[i] < 0 || lm_kf[i] >= n_kf || (i > 0 && lm_kf[i] < lm_kf[i - 1]))
            return c->fail(NRS_ERR_INVALID, "lm_kf must be non-decreasing and in [0, n_kf)");
    }
    for (int64_t i = 0; i < 2 * (int64_t)n_sp; ++i)
        if (sp_ij[i] < 0 || sp_ij[i] >= n_lm) return c->fail(NRS_ERR_INVALID, "spring index out of range");
    for (int64_t i = 0; i < 4 * (int64_t)n_dm; ++i)
        if (dm_idx[i] < 0 || dm_idx[i] >= n_lm) return c->fail(NRS_ERR_INVALID, "damper index out of range");
    NRS_HIP(c, hipSetDevice(c->device));
    dba_free(c);
    DbaProblem* pb = new (std::nothrow) DbaProblem();
    if (!pb) return c->fail(NRS_ERR_ALLOC, "out of host memory");
    c->dba = pb;
    DbaDev& d = pb->d;
    memset(&d, 0, sizeof(d));
    int T = 2;                       // lanes per row; 2 measured best on C2 (profiles/README.md)
    if (const char* e = getenv("NRS_SELL_T")) {
        const int v = atoi(e);
        if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) T = v;
    }
    d.T = T;
    d.K = n_kf;
    d.M = n_lm;
    d.n_sp = n_sp;
    d.n_dm = n_dm;
    d.cam.model = cam->model;
    for (int i = 0; i < 8; ++i) d.cam.p[i] = cam->params[i];
    // constants: reference g2o_optimization.cc:958-973 (float arithmetic, widened)
    const float th2 = sqrtf(5.99f), th3 = sqrtf(0.584f);
    const float sigma_rep = 0.5f, sigma_pos = 0.1f;
    const float sigma_spatial = (float)(0.1 * (double)scale);
    d.info_reproj = (double)(1.0f / (sigma_rep * sigma_rep));
    d.delta_reproj = (double)th2;
    d.info_pos = (double)(1.0f / (sigma_pos * sigma_pos));
    d.delta_pos = 0.0;                                   // no robust kernel on the BA springs (OPT:1057-1071)
    d.info_spatial = (double)(1.0f / (sigma_spatial * sigma_spatial));
    d.delta_spatial = (double)th3;
    d.k_spring = (double)1.1f;
    d.spring_form = 0;

    // ---- row layout: keyframe-major, each keyframe padded to ROW_ALIGN rows
    pb->kf_ptr.assign(n_kf + 1, 0);
    for (int i = 0; i < n_lm; ++i) pb->kf_ptr[lm_kf[i] + 1]++;
    for (int k = 0; k < n_kf; ++k) pb->kf_ptr[k + 1] += pb->kf_ptr[k];
    std::vector<int> kf_grp_ptr(n_kf + 1, 0), grp_kf;
    for (int k = 0; k < n_kf; ++k) {
        const int n = pb->kf_ptr[k + 1] - pb->kf_ptr[k];
        const int ng = std::max(1, (n + ROW_ALIGN - 1) / ROW_ALIGN);
        kf_grp_ptr[k + 1] = kf_grp_ptr[k] + ng;
        for (int g = 0; g < ng; ++g) grp_kf.push_back(k);
    }
    d.n_groups = kf_grp_ptr[n_kf];
    d.n_rows = d.n_groups * ROW_ALIGN;
    const int rows_per_regblk = BLK / T;
    d.n_regblk = d.n_rows / rows_per_regblk;
    d.n_vecblk = d.n_rows / BLK;
    // Inside a keyframe the rows are ordered along a Morton curve of the initial positions, so
    // that the graph neighbours of one workgroup's rows share cache lines (private row order:
    // the C ABI keeps the caller's landmark order).
    pb->lm_row.resize(n_lm);
    std::vector<int> row_lm(d.n_rows, -1);
    {
        float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
        for (int l = 0; l < n_lm; ++l)
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], lm_xyz[3 * l + a]); hi[a] = std::max(hi[a], lm_xyz[3 * l + a]); }
        auto spread = [](uint64_t v) {            // 21 bits -> every third bit
            v &= 0x1fffff;
            v = (v | v << 32) & 0x1f00000000ffffULL;
            v = (v | v << 16) & 0x1f0000ff0000ffULL;
            v = (v | v << 8) & 0x100f00f00f00f00fULL;
            v = (v | v << 4) & 0x10c30c30c30c30c3ULL;
            v = (v | v << 2) & 0x1249249249249249ULL;
            return v;
        };
        const bool morton = getenv("NRS_NO_MORTON") == nullptr;
        std::vector<std::pair<uint64_t, int>> keys;
        for (int k = 0; k < n_kf; ++k) {
            keys.clear();
            for (int l = pb->kf_ptr[k]; l < pb->kf_ptr[k + 1]; ++l) {
                uint64_t code = 0;
                if (morton)
                    for (int a = 0; a < 3; ++a) {
                        const float ext = hi[a] - lo[a];
                        const double f = ext > 0 ? (lm_xyz[3 * l + a] - lo[a]) / ext : 0.0;
                        code |= spread((uint64_t)(f * 2097151.0)) << a;
                    }
                keys.emplace_back(code, l);
            }
            std::stable_sort(keys.begin(), keys.end());
            for (size_t i = 0; i < keys.size(); ++i) {
                const int row = kf_grp_ptr[k] * ROW_ALIGN + (int)i;
                pb->lm_row[keys[i].second] = row;
                row_lm[row] = keys[i].second;
            }
        }
    }
    // ---- incidence lists
    std::vector<std::vector<Inc>> rs(d.n_rows), rd(d.n_rows);
    for (int e = 0; e < n_sp; ++e) {
        const int a = pb->lm_row[sp_ij[2 * e]], b = pb->lm_row[sp_ij[2 * e + 1]];
        Inc i1;
        i1.other[0] = b; i1.other[1] = i1.other[2] = -1; i1.w = sp_d0[e]; i1.meta = (e & 0x3fffffff) | (1 << 30);
        rs[a].push_back(i1);
        Inc i2 = i1;
        i2.other[0] = a; i2.meta = (e & 0x3fffffff);
        rs[b].push_back(i2);
    }
    for (int e = 0; e < n_dm; ++e) {
        int r4[4];
        for (int k = 0; k < 4; ++k) r4[k] = pb->lm_row[dm_idx[4 * e + k]];
        for (int role = 0; role < 4; ++role) {
            Inc in;
            int q = 0;
            for (int k = 0; k < 4; ++k)
                if (k != role) in.other[q++] = r4[k];
            in.w = dm_w[e];
            in.meta = role | ((role == 0 ? 1 : 0) << 2);
            rd[r4[role]].push_back(in);
        }
    }
    std::vector<int> ss_ptr, sd_ptr;
    std::vector<Inc> ss, sd;
    sell_pack(rs, T, ss_ptr, ss);
    sell_pack(rd, T, sd_ptr, sd);
    d.ss.nnz = (int)ss.size();
    d.sd.nnz = (int)sd.size();
    std::vector<int> s_other(ss.size()), s_meta(ss.size()), d_o0(sd.size()), d_o1(sd.size()), d_o2(sd.size()), d_meta(sd.size());
    std::vector<float> s_d0(ss.size()), d_w(sd.size());
    for (size_t i = 0; i < ss.size(); ++i) { s_other[i] = ss[i].other[0]; s_d0[i] = ss[i].w; s_meta[i] = ss[i].meta < 0 ? 0 : ss[i].meta; }
    for (size_t i = 0; i < sd.size(); ++i) { d_o0[i] = sd[i].other[0]; d_o1[i] = sd[i].other[1]; d_o2[i] = sd[i].other[2]; d_w[i] = sd[i].w; d_meta[i] = sd[i].meta; }

    // ---- uploads
    NRS_TRY(dev_upload(c, pb, &d.grp_kf, grp_kf));
    NRS_TRY(dev_upload(c, pb, &d.kf_grp_ptr, kf_grp_ptr));
    NRS_TRY(dev_upload(c, pb, &d.row_lm, row_lm));
    std::vector<float> uv((size_t)d.n_rows * 2, 0.f);
    std::vector<double> xl((size_t)d.n_rows * 3, 0.0);
    for (int l = 0; l < n_lm; ++l) {
        const int row = pb->lm_row[l];
        uv[2 * (size_t)row] = lm_uv[2 * l];
        uv[2 * (size_t)row + 1] = lm_uv[2 * l + 1];
        for (int k = 0; k < 3; ++k) xl[3 * (size_t)row + k] = (double)lm_xyz[3 * l + k];   // OPT:943 cast<double>
    }
    NRS_TRY(dev_upload(c, pb, &d.uv, uv));
    NRS_TRY(dev_upload(c, pb, &d.xl_init, xl));
    std::vector<Pose> poses(n_kf);
    for (int k = 0; k < n_kf; ++k) {
        for (int i = 0; i < 4; ++i) poses[k].q[i] = poses_qt[7 * k + i];
        for (int i = 0; i < 3; ++i) poses[k].t[i] = poses_qt[7 * k + 4 + i];
        quat_normalize(poses[k].q);
    }
    NRS_TRY(dev_upload(c, pb, &d.pose_init, poses));
    NRS_TRY(dev_upload(c, pb, &d.ss.slice_ptr, ss_ptr));
    NRS_TRY(dev_upload(c, pb, &d.sd.slice_ptr, sd_ptr));
    NRS_TRY(dev_upload(c, pb, &d.s_other, s_other));
    NRS_TRY(dev_upload(c, pb, &d.s_d0, s_d0));
    NRS_TRY(dev_upload(c, pb, &d.s_meta, s_meta));
    NRS_TRY(dev_upload(c, pb, &d.d_o0, d_o0));
    NRS_TRY(dev_upload(c, pb, &d.d_o1, d_o1));
    NRS_TRY(dev_upload(c, pb, &d.d_o2, d_o2));
    NRS_TRY(dev_upload(c, pb, &d.d_w, d_w));
    NRS_TRY(dev_upload(c, pb, &d.d_meta, d_meta));
    const size_t nr = (size_t)d.n_rows, K = (size_t)n_kf;
    for (int s = 0; s < 2; ++s) {
        NRS_TRY(dev_alloc(c, pb, (void**)&d.pose[s], sizeof(Pose) * K));
        NRS_TRY(dev_alloc(c, pb, (void**)&d.xl[s], sizeof(double) * 3 * nr));
    }
    NRS_TRY(dev_alloc(c, pb, (void**)&d.D, sizeof(double) * 6 * nr));
    NRS_TRY(dev_alloc(c, pb, (void**)&d.Hpl, sizeof(double) * 18 * nr));
    NRS_TRY(dev_alloc(c, pb, (void**)&d.s_g, sizeof(double) * 3 * ss.size()));
    NRS_TRY(dev_alloc(c, pb, (void**)&d.d_s, sizeof(double) * sd.size()));
    NRS_TRY(dev_alloc(c, pb, (void**)&d.Hpp, sizeof(double) * 21 * K));
    NRS_TRY(dev_alloc(c, pb, (void**)&d.bp, sizeof(double) * 6 * K));
    NRS_TRY(dev_alloc(c, pb, (void**)&d.bl, sizeof(double) * 3 * nr));
    NRS_TRY(dev_alloc(c, pb, (void**)&d.Dinv, sizeof(double) * 6 * nr));
    NRS_TRY(dev_alloc(c, pb, (void**)&d.Hppinv, sizeof(double) * 36 * K));
    double** pv[] = {&d.xp, &d.rp, &d.up, &d.pp, &d.sp, &d.wp};
    for (auto p : pv) NRS_TRY(dev_alloc(c, pb, (void**)p, sizeof(double) * 6 * K));
    double** rvv[] = {&d.xv, &d.rv, &d.uv3, &d.pv, &d.sv, &d.wv};
    for (auto p : rvv) NRS_TRY(dev_alloc(c, pb, (void**)p, sizeof(double) * 3 * nr));
    const int n_vecblk = (d.n_rows + BLK - 1) / BLK;
    NRS_TRY(dev_alloc(c, pb, (void**)&d.part_lin, sizeof(double) * 32 * (size_t)d.n_groups));
    NRS_TRY(dev_alloc(c, pb, (void**)&d.part_reg, sizeof(double) * 2 * (size_t)d.n_regblk));
    NRS_TRY(dev_alloc(c, pb, (void**)&d.part_spmv, sizeof(double) * NPART * (size_t)d.n_regblk));
    NRS_TRY(dev_alloc(c, pb, (void**)&d.part_apply, sizeof(double) * (size_t)std::max(d.n_regblk, n_vecblk)));
    NRS_TRY(dev_alloc(c, pb, (void**)&d.scal, sizeof(double) * SC_N));
    NRS_TRY(dev_alloc(c, pb, (void**)&d.flags, sizeof(int) * 8));
    NRS_HIP(c, hipMemsetAsync(d.part_apply, 0, sizeof(double) * (size_t)std::max(d.n_regblk, n_vecblk), c->stream));
    NRS_HIP(c, hipMemsetAsync(d.scal, 0, sizeof(double) * SC_N, c->stream));
    NRS_HIP(c, hipMemsetAsync(d.flags, 0, sizeof(int) * 8, c->stream));
    NRS_HIP(c, hipHostMalloc((void**)&pb->h_scal, sizeof(double) * SC_N));
    NRS_HIP(c, hipHostMalloc((void**)&pb->h_flags, sizeof(int) * 8));
    pb->sp_ij.assign(sp_ij, sp_ij + 2 * (size_t)n_sp);
    pb->sp_d0.assign(sp_d0, sp_d0 + (size_t)n_sp);
    pb->dm_idx.assign(dm_idx, dm_idx + 4 * (size_t)n_dm);
    pb->dm_w.assign(dm_w, dm_w + (size_t)n_dm);
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    return NRS_OK;
}

static int dba_reset_impl(nrs_ctx* c) {
    DbaProblem* pb = c->dba;
    if (!pb) return c->fail(NRS_ERR_STATE, "no BA problem uploaded");
    DbaDev& d = pb->d;
    pb->cur = 0;
    NRS_HIP(c, hipMemcpyAsync(d.pose[0], d.pose_init, sizeof(Pose) * d.K, hipMemcpyDeviceToDevice, c->stream));
    NRS_HIP(c, hipMemcpyAsync(d.xl[0], d.xl_init, sizeof(double) * 3 * (size_t)d.n_rows, hipMemcpyDeviceToDevice, c->stream));
    NRS_HIP(c, hipMemcpyAsync(d.xl[1], d.xl_init, sizeof(double) * 3 * (size_t)d.n_rows, hipMemcpyDeviceToDevice, c->stream));
    return NRS_OK;
}

struct Timer {                       // HIP-event timing of one launch when profiling is on
    nrs_ctx* c;
    double* acc;
    int64_t* cnt;
    Timer(nrs_ctx* c_, double* a, int64_t* n) : c(c_), acc(a), cnt(n) {
        if (c->opt.profile) (void)hipEventRecord(c->ev0, c->stream);
    }
    ~Timer() {
        if (c->opt.profile) {
            (void)hipEventRecord(c->ev1, c->stream);
            (void)hipEventSynchronize(c->ev1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
            *acc += ms;
            *cnt += 1;
        }
    }
};

template <bool LIN>
static void launch_reg(nrs_ctx* c, const DbaDev& d, const double* xl) {
    const dim3 g(((d.n_regblk + 7) / 8) * 8), b(BLK);
    switch (d.T) {
        case 1: hipLaunchKernelGGL((k_reg<1, LIN>), g, b, 0, c->stream, d, xl); break;
        case 8: hipLaunchKernelGGL((k_reg<8, LIN>), g, b, 0, c->stream, d, xl); break;
        case 16: hipLaunchKernelGGL((k_reg<16, LIN>), g, b, 0, c->stream, d, xl); break;
        case 4: hipLaunchKernelGGL((k_reg<4, LIN>), g, b, 0, c->stream, d, xl); break;
        default: hipLaunchKernelGGL((k_reg<2, LIN>), g, b, 0, c->stream, d, xl); break;
    }
}

static void launch_spmv(nrs_ctx* c, const DbaDev& d, double lam) {
    const dim3 g(((d.n_regblk + 7) / 8) * 8), b(BLK);
    switch (d.T) {
        case 1: hipLaunchKernelGGL((k_spmv<1>), g, b, 0, c->stream, d, lam); break;
        case 8: hipLaunchKernelGGL((k_spmv<8>), g, b, 0, c->stream, d, lam); break;
        case 16: hipLaunchKernelGGL((k_spmv<16>), g, b, 0, c->stream, d, lam); break;
        case 4: hipLaunchKernelGGL((k_spmv<4>), g, b, 0, c->stream, d, lam); break;
        default: hipLaunchKernelGGL((k_spmv<2>), g, b, 0, c->stream, d, lam); break;
    }
}

// errors (+ linearisation) at a given state; leaves chi2 (and max diag) in scal[]
template <bool LIN>
static int evaluate(nrs_ctx* c, DbaProblem* pb, int which) {
    const DbaDev& d = pb->d;
    const dim3 gg(((d.n_groups + 7) / 8) * 8), b(BLK);
    if (LIN) {
        Timer t(c, &c->prof.linearize_ms, &c->prof.linearize_launches);
        hipLaunchKernelGGL((k_reproj<LIN>), gg, b, 0, c->stream, d, d.pose[which], d.xl[which]);
        launch_reg<LIN>(c, d, d.xl[which]);
    } else {
        hipLaunchKernelGGL((k_reproj<LIN>), gg, b, 0, c->stream, d, d.pose[which], d.xl[which]);
        launch_reg<LIN>(c, d, d.xl[which]);
    }
    hipLaunchKernelGGL((k_finalize<LIN>), dim3(1), b, 0, c->stream, d);
    NRS_HIP(c, hipGetLastError());
    return NRS_OK;
}

static int read_scalars(nrs_ctx* c, DbaProblem* pb) {
    NRS_HIP(c, hipMemcpyAsync(pb->h_scal, pb->d.scal, sizeof(double) * SC_N, hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipMemcpyAsync(pb->h_flags, pb->d.flags, sizeof(int) * 8, hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    return NRS_OK;
}

// (H + lam I) x = b by block-Jacobi PCG; returns iterations, ok=false on non-finite values
static int pcg_solve(nrs_ctx* c, DbaProblem* pb, double lam, int* iters, bool* ok) {
    const DbaDev& d = pb->d;
    const int n_vecblk = (d.n_rows + BLK - 1) / BLK;
    const int n_poseblk = (d.K + 41) / 42;
    const double tol2 = c->opt.pcg_rtol * c->opt.pcg_rtol;
    NRS_HIP(c, hipMemsetAsync(d.flags, 0, sizeof(int) * 8, c->stream));
    hipLaunchKernelGGL(k_trial_setup, dim3(n_vecblk), dim3(BLK), 0, c->stream, d, lam);
    int it = 0;
    while (true) {
        const int stop = std::min(it + c->opt.pcg_batch, c->opt.pcg_max_iters);
        for (; it < stop; ++it) {
            {
                Timer t(c, &c->prof.spmv_ms, &c->prof.spmv_launches);
                launch_spmv(c, d, lam);
            }
            {
                Timer t(c, &c->prof.vec_ms, &c->prof.vec_launches);
                hipLaunchKernelGGL(k_pcg_update, dim3(((n_vecblk + 7) / 8) * 8 + n_poseblk), dim3(BLK), 0, c->stream, d, lam, it, tol2, n_vecblk);
            }
        }
        NRS_HIP(c, hipGetLastError());
        NRS_HIP(c, hipMemcpyAsync(pb->h_flags, d.flags, sizeof(int) * 8, hipMemcpyDeviceToHost, c->stream));
        NRS_HIP(c, hipStreamSynchronize(c->stream));
        if (pb->h_flags[0] || it >= c->opt.pcg_max_iters) break;
    }
    *iters = pb->h_flags[1];
    *ok = pb->h_flags[2] == 0;
    return NRS_OK;
}

// g2o SparseOptimizer::optimize + OptimizationAlgorithmLevenberg::solve on the resident problem
static int dba_optimize_impl(nrs_ctx* c, int iters, nrs_lm_trace* trace) {
    DbaProblem* pb = c->dba;
    if (!pb) return c->fail(NRS_ERR_STATE, "no BA problem uploaded");
    if (iters < 0) return c->fail(NRS_ERR_INVALID, "iters < 0");
    NRS_HIP(c, hipSetDevice(c->device));
    DbaDev& d = pb->d;
    const int n_vecblk = (d.n_rows + BLK - 1) / BLK;
    double lam = -1, ni = 2;
    int ntr = 0, done = 0;
    if (trace) { trace->count = 0; trace->iterations = 0; }
    for (int it = 0; it < iters; ++it) {
        NRS_TRY(evaluate<true>(c, pb, pb->cur));
        NRS_TRY(read_scalars(c, pb));
        double chi = pb->h_scal[SC_CHI];
        if (it == 0) { lam = 1e-5 * pb->h_scal[SC_MAXDIAG]; ni = 2; }
        if (!std::isfinite(chi) || !std::isfinite(lam)) return c->fail(NRS_ERR_NUMERIC, "non-finite chi2/lambda at LM iteration %d", it);
        double rho = 0;
        int qmax = 0;
        do {
            int inner = 0;
            bool ok = true;
            NRS_TRY(pcg_solve(c, pb, lam, &inner, &ok));
            const int trial = 1 - pb->cur;
            {
                Timer t(c, &c->prof.update_ms, &c->prof.update_launches);
                hipLaunchKernelGGL(k_apply, dim3(n_vecblk), dim3(BLK), 0, c->stream, d, lam, d.pose[pb->cur], d.xl[pb->cur], d.pose[trial], d.xl[trial]);
            }
            NRS_TRY(evaluate<false>(c, pb, trial));
            NRS_TRY(read_scalars(c, pb));
            const double temp = ok ? pb->h_scal[SC_CHI] : 1.7976931348623157e308;
            const double scale = pb->h_scal[SC_SCALE] + 1e-3;
            rho = (chi - temp) / scale;
            const bool accepted = rho > 0 && std::isfinite(temp);
            if (trace) {
                if (trace->trials && ntr < trace->capacity) {
                    nrs_lm_trial& T = trace->trials[ntr];
                    T.round = 0; T.iter = it; T.trial = qmax; T.accepted = accepted; T.solver_ok = ok;
                    T.inner_iters = inner; T.lambda = lam; T.chi2 = chi; T.chi2_new = temp; T.rho = rho;
                }
            }
            ++ntr;
            if (accepted) {
                double alpha = 1.0 - std::pow(2 * rho - 1, 3);
                alpha = std::min(alpha, 2.0 / 3.0);
                lam *= std::max(1.0 / 3.0, alpha);
                ni = 2;
                chi = temp;
                pb->cur = trial;                       // discardTop: the trial state becomes current
            } else {
                lam *= ni;
                ni *= 2;                               // pop: current state untouched
                if (!std::isfinite(lam)) break;
            }
            ++qmax;
        } while (rho < 0 && qmax < 10);
        ++done;
        if (qmax == 10 || rho == 0 || !std::isfinite(lam)) break;
    }
    if (trace) { trace->count = ntr; trace->iterations = done; }
    return NRS_OK;
}

static int dba_download_impl(nrs_ctx* c, double* poses_qt, double* lm_xyz) {
    DbaProblem* pb = c->dba;
    if (!pb) return c->fail(NRS_ERR_STATE, "no BA problem uploaded");
    DbaDev& d = pb->d;
    std::vector<Pose> poses(d.K);
    std::vector<double> xl((size_t)d.n_rows * 3);
    NRS_HIP(c, hipMemcpyAsync(poses.data(), d.pose[pb->cur], sizeof(Pose) * d.K, hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipMemcpyAsync(xl.data(), d.xl[pb->cur], sizeof(double) * xl.size(), hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    if (poses_qt)
        for (int k = 0; k < d.K; ++k) {
            for (int i = 0; i < 4; ++i) poses_qt[7 * k + i] = poses[k].q[i];
            for (int i = 0; i < 3; ++i) poses_qt[7 * k + 4 + i] = poses[k].t[i];
        }
    if (lm_xyz)
        for (int l = 0; l < d.M; ++l)
            for (int k = 0; k < 3; ++k) lm_xyz[3 * (size_t)l + k] = xl[3 * (size_t)pb->lm_row[l] + k];
    return NRS_OK;
}

}  // namespace nrs

using namespace nrs;

extern "C" int nrs_dba_upload(nrs_ctx* c, const nrs_camera* cam, int32_t n_kf, const double* poses_qt,
                              int32_t n_lm, const float* lm_xyz, const int32_t* lm_kf, const float* lm_uv,
                              int32_t n_sp, const int32_t* sp_ij, const float* sp_d0,
                              int32_t n_dm, const int32_t* dm_idx, const float* dm_w, float scale) {
    if (!c) return NRS_ERR_INVALID;
    int rc = dba_upload_impl(c, cam, n_kf, poses_qt, n_lm, lm_xyz, lm_kf, lm_uv, n_sp, sp_ij, sp_d0, n_dm, dm_idx, dm_w, scale);
    if (rc != NRS_OK) { dba_free(c); return rc; }
    return dba_reset_impl(c);
}

extern "C" int nrs_dba_reset(nrs_ctx* c) { return c ? dba_reset_impl(c) : NRS_ERR_INVALID; }

extern "C" int nrs_dba_optimize(nrs_ctx* c, int32_t iters, nrs_lm_trace* trace) {
    return c ? dba_optimize_impl(c, iters, trace) : NRS_ERR_INVALID;
}

extern "C" int nrs_dba_download(nrs_ctx* c, double* poses_qt, double* lm_xyz) {
    return c ? dba_download_impl(c, poses_qt, lm_xyz) : NRS_ERR_INVALID;
}

extern "C" int nrs_dba_solve(nrs_ctx* c, const nrs_camera* cam, int32_t n_kf, double* poses_qt,
                             int32_t n_lm, float* lm_xyz, const int32_t* lm_kf, const float* lm_uv,
                             int32_t n_sp, const int32_t* sp_ij, const float* sp_d0,
                             int32_t n_dm, const int32_t* dm_idx, const float* dm_w,
                             float scale, int32_t iters, nrs_lm_trace* trace) {
    if (!c) return NRS_ERR_INVALID;
    NRS_TRY(nrs_dba_upload(c, cam, n_kf, poses_qt, n_lm, lm_xyz, lm_kf, lm_uv, n_sp, sp_ij, sp_d0, n_dm, dm_idx, dm_w, scale));
    NRS_TRY(dba_optimize_impl(c, iters, trace));
    std::vector<double> xyz((size_t)n_lm * 3);
    NRS_TRY(dba_download_impl(c, poses_qt, xyz.data()));
    for (size_t i = 0; i < xyz.size(); ++i) lm_xyz[i] = (float)xyz[i];      // OPT:1158 cast<float>
    return NRS_OK;
}

extern "C" int nrs_dba_residuals(nrs_ctx* c, double* r_reproj, double* r_spring, double* r_damper) {
    if (!c) return NRS_ERR_INVALID;
    DbaProblem* pb = c->dba;
    if (!pb) return c->fail(NRS_ERR_STATE, "no BA problem uploaded");
    if (!r_reproj || !r_spring || !r_damper) return c->fail(NRS_ERR_INVALID, "null output");
    DbaDev& d = pb->d;
    int *lm_row, *sp, *dm;
    float *d0, *w;
    double *rr, *rs, *rd;
    std::vector<void*> tmp;
    auto al = [&](void** p, size_t bytes) { hipError_t e = hipMalloc(p, bytes ? bytes : 16); if (e == hipSuccess) tmp.push_back(*p); return e; };
    NRS_HIP(c, al((void**)&lm_row, sizeof(int) * d.M));
    NRS_HIP(c, al((void**)&sp, sizeof(int) * 2 * (size_t)d.n_sp));
    NRS_HIP(c, al((void**)&dm, sizeof(int) * 4 * (size_t)d.n_dm));
    NRS_HIP(c, al((void**)&d0, sizeof(float) * (size_t)d.n_sp));
    NRS_HIP(c, al((void**)&w, sizeof(float) * (size_t)d.n_dm));
    NRS_HIP(c, al((void**)&rr, sizeof(double) * 2 * (size_t)d.M));
    NRS_HIP(c, al((void**)&rs, sizeof(double) * (size_t)d.n_sp));
    NRS_HIP(c, al((void**)&rd, sizeof(double) * 3 * (size_t)d.n_dm));
    NRS_HIP(c, hipMemcpy(lm_row, pb->lm_row.data(), sizeof(int) * d.M, hipMemcpyHostToDevice));
    if (d.n_sp) {
        NRS_HIP(c, hipMemcpy(sp, pb->sp_ij.data(), sizeof(int) * 2 * (size_t)d.n_sp, hipMemcpyHostToDevice));
        NRS_HIP(c, hipMemcpy(d0, pb->sp_d0.data(), sizeof(float) * (size_t)d.n_sp, hipMemcpyHostToDevice));
    }
    if (d.n_dm) {
        NRS_HIP(c, hipMemcpy(dm, pb->dm_idx.data(), sizeof(int) * 4 * (size_t)d.n_dm, hipMemcpyHostToDevice));
        NRS_HIP(c, hipMemcpy(w, pb->dm_w.data(), sizeof(float) * (size_t)d.n_dm, hipMemcpyHostToDevice));
    }
    const int n = std::max(d.M, std::max(d.n_sp, d.n_dm));
    hipLaunchKernelGGL(k_tap_residuals, dim3((n + 255) / 256), dim3(256), 0, c->stream, d, d.pose[pb->cur], d.xl[pb->cur], lm_row, sp, d0, dm, w, rr, rs, rd);
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    NRS_HIP(c, hipMemcpy(r_reproj, rr, sizeof(double) * 2 * (size_t)d.M, hipMemcpyDeviceToHost));
    if (d.n_sp) NRS_HIP(c, hipMemcpy(r_spring, rs, sizeof(double) * (size_t)d.n_sp, hipMemcpyDeviceToHost));
    if (d.n_dm) NRS_HIP(c, hipMemcpy(r_damper, rd, sizeof(double) * 3 * (size_t)d.n_dm, hipMemcpyDeviceToHost));
    for (void* p : tmp) (void)hipFree(p);
    return NRS_OK;
}

extern "C" int nrs_dba_gradient(nrs_ctx* c, double* b, double* diag) {
    if (!c) return NRS_ERR_INVALID;
    DbaProblem* pb = c->dba;
    if (!pb) return c->fail(NRS_ERR_STATE, "no BA problem uploaded");
    if (!b || !diag) return c->fail(NRS_ERR_INVALID, "null output");
    DbaDev& d = pb->d;
    NRS_TRY(evaluate<true>(c, pb, pb->cur));
    std::vector<double> bp(6 * (size_t)d.K), Hpp(21 * (size_t)d.K), bl(3 * (size_t)d.n_rows), D(6 * (size_t)d.n_rows);
    NRS_HIP(c, hipMemcpyAsync(bp.data(), d.bp, sizeof(double) * bp.size(), hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipMemcpyAsync(Hpp.data(), d.Hpp, sizeof(double) * Hpp.size(), hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipMemcpyAsync(bl.data(), d.bl, sizeof(double) * bl.size(), hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipMemcpyAsync(D.data(), d.D, sizeof(double) * D.size(), hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    static const int dg[6] = {0, 6, 11, 15, 18, 20};
    for (int k = 0; k < d.K; ++k)
        for (int a = 0; a < 6; ++a) { b[6 * k + a] = bp[6 * k + a]; diag[6 * k + a] = Hpp[21 * k + dg[a]]; }
    static const int d3[3] = {0, 3, 5};
    for (int l = 0; l < d.M; ++l) {
        const size_t row = (size_t)pb->lm_row[l];
        for (int a = 0; a < 3; ++a) {
            b[6 * (size_t)d.K + 3 * (size_t)l + a] = bl[3 * row + a];
            diag[6 * (size_t)d.K + 3 * (size_t)l + a] = D[6 * row + d3[a]];
        }
    }
    return NRS_OK;
}
