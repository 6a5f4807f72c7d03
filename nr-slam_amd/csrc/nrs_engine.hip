// Graph Levenberg-Marquardt engine for MI355X: the numerical core behind
//   a2 CameraPoseAndDeformationOptimization (reference g2o_optimization.cc:148-557) and
//   a3 LocalDeformableBundleAdjustment       (reference g2o_optimization.cc:880-1161),
// i.e. the g2o machinery those functions drive (third_party/g2o/g2o/core/
// optimization_algorithm_levenberg.cpp:57-174, block_solver.hpp:495-562,
// base_fixed_sized_edge.hpp:49-133, robust_kernel_impl.cpp:60-74, sparse_optimizer.cpp:203-285).
//
// Design (MI355X-first, not a translation of g2o's pointer graph):
//   * unknowns: K pose blocks (6) then one 3-dof block per landmark ROW.  Rows are laid out
//     pose-major and every pose's rows are padded to a multiple of ROW_ALIGN, so a workgroup
//     never straddles two poses and the per-pose reductions are plain block reductions.  Inside
//     a pose the rows follow a Morton curve of the initial positions: graph neighbours share
//     cache lines.
//   * every edge is owned by the rows it touches ("incidence lists", sliced-ELL layout: a wave64
//     serves 64/T rows with T lanes per row; element (row r, lane t, step j) of a slice lives at
//     base + j*64 + r*T + t so all loads of a wave are one contiguous 64-element run).  Nothing
//     is scattered: no atomics, no assembly maps, bit-reproducible sums.
//   * H is never assembled, not even block-wise.  (H + lambda I) x = b is solved by preconditioned
//     conjugate gradients (single-reduction Chronopoulos-Gear form) whose operator is applied in
//     FACTORED form from 16-byte incidence records and the LDS-staged linearisation point:
//         spring  : block = qc v v^T, v = x_i - x_j re-formed from staged positions; stores qc  (8 B)
//         damper  : all 16 blocks are +-s I3; stores s                                          (8 B)
//         reproj  : J^T w J with J rebuilt from the fp32 projection Jacobian kept per row       (32 B)
//     A tile (256/T rows) and the rows its incidences reach (the halo) are staged into LDS once per
//     launch; neighbour ids in the records are tile-local.  Large problems: two launches per
//     iteration (k_spmv_f + k_pcg_update).  Small problems (< 32768 rows): ONE launch per iteration
//     (k_pcg_fused: vector update of iteration k-1 + operator of iteration k, halo rows re-derived
//     locally), with, for single-pose problems, a two-level preconditioner whose coarse solve is
//     applied inside that launch (k_coarse_tile / k_coarse_reduce / k_coarse_invert).
//     The reference factorises the same matrix with a sparse Cholesky (no Schur: H_ll is not
//     block diagonal); PCG to 1e-10 relative residual reproduces its iterates.
//   * g2o levels / fixed vertices are byte masks: inactive edges store zero factors, fixed rows
//     are identity rows whose columns vanish because their PCG vectors stay zero.
//   * LM control flow (lambda schedule, accept/reject, <=10 trials) runs on the host exactly as in
//     g2o.  A trial's first PCG batch, the state update and the chi2 evaluation go out in one
//     enqueue and are read back with one synchronisation (results are written by the last kernel
//     into mapped host memory); trials that are going to be rejected are recognised at the 1e-1 ..
//     1e-4 milestones of the inner solve and not solved further (nrs_options.exact_trials = 1
//     turns that off); the iterates are the reference's either way.
//   * XCD-aware launch order: logical tile = (blockIdx % 8) * ceil(nb/8) + blockIdx / 8, so each
//     XCD's L2 serves one contiguous run of rows (neighbour gathers stay inside it).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <new>
#include <thread>
#include "nrs_engine.hpp"

namespace nrs {

constexpr int ROW_ALIGN = 256;       // pose row padding; also rows per k_reproj workgroup
constexpr int BLK = 256;             // threads per workgroup everywhere
constexpr int NPART = 12;            // per-block partial slots of the SpMV kernel: [0..2] dots, [3..8] pose sums
constexpr int CO_MAX = 84;           // largest coarse system of the two-level preconditioner (fits one workgroup's LDS)
constexpr int CO_GMAX = (CO_MAX - 6) / 3;   // row groups of the coarse level
static_assert(CO_GMAX <= 32, "k_coarse_tile keeps the reached groups in a 32-bit mask");

// incidence meta bits
constexpr int SM_COUNT = 1 << 30;    // spring: this incidence adds the edge's rho to chi2
constexpr int SM_ACTIVE = 1 << 29;   // spring: edge is at level 0
constexpr int DM_COUNT = 1 << 2;     // damper: bits 0-1 role
constexpr int DM_ACTIVE = 1 << 3;
constexpr int DM_UNARY = 1 << 4;     // damper: the other vertex is a value, not a variable

// packed incidence records of the LDS-staged path: one 16-byte load per incidence; neighbour ids
// are tile-local (own rows, then halo).  The operator is applied in factored form: a spring block is
// qc * v v^T with v = x_i - x_j re-formed from the staged linearisation point (qc = rho' Omega cg^2),
// a reprojection block is J^T w J with J rebuilt from the fp32 projection Jacobian kept per row.
struct __attribute__((aligned(16))) SpringRec { double qc; uint16_t other, meta; float d0; };   // 16 B
struct __attribute__((aligned(16))) RowRec { float J[6]; double w; };                           // 32 B
struct __attribute__((aligned(16))) DamperRec { uint16_t o0, o1, o2, meta; double s; };                  // 16 B
constexpr uint16_t REC_NONE = 0xFFFF;
constexpr uint16_t SR_ACTIVE = 1, SR_COUNT = 2;

struct Dev {
    int K, M, n_rows, n_groups;      // poses, vertices, padded rows, ROW_ALIGN groups
    int T;                           // lanes per row
    int n_sp, n_dm, n_un;
    int n_regblk, n_vecblk;
    Cam cam;
    double info_reproj, delta_reproj, info_pos, delta_pos, info_spatial, delta_spatial, k_spring;
    int spring_form;
    // rows
    int* grp_pose;                   // n_groups -> pose index
    int* pose_grp_ptr;               // K+1 -> group ranges
    uint8_t* rflag;                  // n_rows
    uint8_t* pose_fixed;             // K
    float* uv;                       // n_rows x 2
    double* X0;                      // n_rows x 3 or null
    // incidences (sliced ELL)
    int* ss_ptr; int ss_nnz; int* sd_ptr; int sd_nnz;
    int* s_other; float* s_d0; int* s_meta;
    int* d_o0; int* d_o1; int* d_o2; float* d_w; int* d_meta;
    // LDS staging: neighbour ids above are LOCAL to the workgroup's tile: [0, tile_rows) = own rows,
    // tile_rows + i = halo_rows[halo_ptr[b] + i]
    int use_lds, tile_rows, max_halo;
    int max_halo_s;                  // halo lists start with the spring (same-keyframe) neighbours: at most this many
    // tiles come in two classes so that a few tiles with very large halos do not set the LDS size
    // (= occupancy) of all: tile_list = class-0 tiles, then class-1 tiles; caps per class
    int* tile_list; int n_tiles_cls[2]; int cap_h[2], cap_s[2];
    int* halo_ptr; int* halo_rows; int* halo_ns;   // halo_ns[b] = number of spring-halo rows of tile b
    SpringRec* s_rec; DamperRec* d_rec;
    RowRec* rowrec;                  // n_rows (LDS path): reprojection factors of the linearisation point
    Pose* lin_pose; double* lin_xl;  // the linearisation point itself (= pose[cur], xl[cur])
    // state (two copies: current / trial, swapped on accept)
    Pose* pose[2]; double* xl[2];
    Pose* pose_init; double* xl_init;
    // linearisation
    double* D;                       // n_rows x 6   (xx xy xz yy yz zz)
    double* Hpl;                     // 18 x n_rows  (component-major; gather fallback path only)
    double* s_g;                     // 3 x nnz_s
    double* d_s;                     // nnz_d
    double* Hpp;                     // K x 21
    double* bp; double* bl;          // 6K, 3 n_rows
    double* Dinv; double* Hppinv;    // n_rows x 6, K x 36
    // PCG vectors: pose part [6K] and row part [3 n_rows]
    double *xp, *rp, *up, *pp, *sp, *wp;
    double *xv, *rv, *uv3, *pv, *sv, *wv;
    // second halves of the ping-pong pairs used by the fused small-problem iteration
    double *rp2, *sp2, *up2, *rv2, *sv2, *wv2, *part_spmv2;
    int fused;
    // two-level preconditioner of the fused path (single pose): coarse unknowns = one translation per
    // 256-row group + the pose; M^-1 = block-Jacobi + Z (Z^T H Z + lambda Z^T Z)^-1 Z^T
    int coarse, co_n;                // enabled, number of coarse unknowns (3 n_groups + 6)
    double* co_ct;                   // n_regblk x n_groups x 6: sum of H_ij over i in tile, j in each row group
    double* co_cp;                   // n_regblk x 18: sum of H_lp over the tile's rows (3x6)
    double* co_tb;                   // n_regblk x 4: sum of b (3) and number of free rows
    double* part_ts; double* part_ts2;   // 9 x n_regblk (component-major, ping-pong): tile sums of r, s, w
    double* co_bt;                   // n_regblk x 6: sum of H_ij over i, j in the tile (tile-level diagonal block)
    double* co_bti;                  // n_regblk x 6: (B_t + lambda n_t I)^-1 of the current trial (0 if not positive)
    double* co_c0;                   // co_n x co_n: Z^T H Z ; co_nn: Z^T Z diagonal ; co_bc: Z^T b
    double* co_nn; double* co_bc;
    double* co_inv;                  // co_n x co_n: (C0 + lambda N)^-1 of the current trial
    double* co_y0;                   // co_n: its product with Z^T b (start vector of the trial)
    int* tile_desc;                  // fused path: 8 ints per tile {pose, first tile of pose, end tile of pose, halo begin, halo count, 0,0,0}
    int* halo_fix;                   // fused path: BLK ints per tile = the first BLK halo rows (fixed stride: no pointer chase)
    // large problems: the SpMV partials are pre-reduced by k_reduce_partials (one launch) instead of
    // being re-summed by every workgroup of the update kernel (which is O(workgroups^2) reads)
    int hier;
    double* red;                     // [0..2] r.u, w.u, cross ; [3 + 6k + a] pose sums
    // partials / scalars
    double* part_lin;                // n_groups x 32   (reproj kernel: 27 pose sums + chi)
    double* part_reg;                // n_regblk x 2    (chi, maxdiag)
    double* part_spmv;               // n_regblk x NPART
    double* part_apply;              // n_vecblk
    double* scal;
    double* h_scal; int* h_flags;     // host-mapped mirrors, written by k_finalize / k_publish (no copy kernels)
    int* flags;                      // [0] pcg done, [1] pcg iterations, [2] nan flag
};

enum { SC_CHI = 0, SC_MAXDIAG = 1, SC_SCALE = 2, SC_GAMMA0 = 3, SC_SLOT0 = 4, SC_SLOT1 = 6, SC_N = 16 };

struct Engine {
    Dev d;
    Arena* arena = nullptr;
    int cur = 0;
    int pred_iters = 0;              // inner iterations of the last fully solved LM trial (sizes later batches)
    int pred_peek = 0;               // iterations the last trial needed to reach the first peek milestone
    bool first_trial_accepted = false;   // outcome of the first trial of the previous LM iteration
    double* h_scal = nullptr;        // pinned host mirrors
    int* h_flags = nullptr;
    std::vector<int> vrow;           // vertex -> row
    // host copies needed to rewrite masks and to run the edge taps
    std::vector<int> sp_ij, dm_idx, un_ij;
    std::vector<float> sp_d0, dm_w, un_w;
    std::vector<int> sp_pos, dm_pos, un_pos;     // SELL positions of every incidence (2 / 4 / 1 per edge)
    std::vector<int> h_s_meta, h_d_meta;
    std::vector<SpringRec> h_s_rec;
    std::vector<DamperRec> h_d_rec;
    std::vector<uint8_t> h_rflag, h_pose_fixed;
    // device copies for the taps
    int *t_vrow = nullptr, *t_sp = nullptr, *t_dm = nullptr;
    float *t_d0 = nullptr, *t_w = nullptr;
    double* t_out = nullptr;
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

// same reduction, totals written to out[0..N) by the first N threads
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

template <int T>
__device__ inline double sub_sum_t(double v) { return group_sum<T>(v); }   // reduce over the T lanes of a row

__device__ inline bool inv3_sym(const double* d /*xx xy xz yy yz zz*/, double lam, double* o) {
    const double a = d[0] + lam, b = d[1], c = d[2], e = d[3] + lam, f = d[4], g = d[5] + lam;
    const double c00 = e * g - f * f, c01 = c * f - b * g, c02 = b * f - c * e;
    const double det = a * c00 + b * c01 + c * c02;
    const double id = 1.0 / det;
    o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
    o[3] = (a * g - c * c) * id; o[4] = (b * c - a * f) * id; o[5] = (a * e - b * b) * id;
    return det > 0;
}

// tile-level coarse correction: y = Bi rc with Bi = (B_t + lambda n_t I)^-1 prepared per trial by k_coarse_invert
__device__ inline void tile_level(const double* Bi /*6*/, const double* rc, double* y) {
    y[0] = Bi[0] * rc[0] + Bi[1] * rc[1] + Bi[2] * rc[2];
    y[1] = Bi[1] * rc[0] + Bi[3] * rc[1] + Bi[4] * rc[2];
    y[2] = Bi[2] * rc[0] + Bi[4] * rc[1] + Bi[5] * rc[2];
}

// stage 3-vectors of the tile's own rows and of its halo rows into LDS (optionally adding X0).
// The halo is a gather through an index list: all indices of a thread are requested first, then all
// rows, so that a thread has its 2-4 gathers in flight together instead of one dependent pair at a time.
constexpr int STAGE_K = 4;
__device__ inline void stage_rows(const Dev& P, int b, int tid, const double* __restrict__ v, const double* __restrict__ add,
                                  double* lds) {
    const int row0 = b * P.tile_rows;
    const int hb = P.halo_ptr[b], hn = P.halo_ptr[b + 1] - hb;
    int idx[STAGE_K];
#pragma unroll
    for (int k = 0; k < STAGE_K; ++k) { const int i = tid + k * BLK; idx[k] = i < hn ? P.halo_rows[hb + i] : -1; }
    for (int i = tid; i < 3 * P.tile_rows; i += BLK) lds[i] = v[3 * (size_t)row0 + i] + (add ? add[3 * (size_t)row0 + i] : 0.0);
    double val[STAGE_K][3];
#pragma unroll
    for (int k = 0; k < STAGE_K; ++k) {
        if (idx[k] >= 0) {
            const size_t r = (size_t)idx[k];
#pragma unroll
            for (int c = 0; c < 3; ++c) val[k][c] = v[3 * r + c] + (add ? add[3 * r + c] : 0.0);
        }
    }
#pragma unroll
    for (int k = 0; k < STAGE_K; ++k) {
        if (idx[k] >= 0) {
            double* d = lds + 3 * (size_t)(P.tile_rows + tid + k * BLK);
            d[0] = val[k][0]; d[1] = val[k][1]; d[2] = val[k][2];
        }
    }
    for (int i = tid + STAGE_K * BLK; i < hn; i += BLK) {         // very large halos
        const size_t r = (size_t)P.halo_rows[hb + i];
        double* d = lds + 3 * (size_t)(P.tile_rows + i);
        d[0] = v[3 * r] + (add ? add[3 * r] : 0.0);
        d[1] = v[3 * r + 1] + (add ? add[3 * r + 1] : 0.0);
        d[2] = v[3 * r + 2] + (add ? add[3 * r + 2] : 0.0);
    }
}

// u and the (spring) positions of the linearisation point, one pass over the halo list
__device__ inline void stage_rows2(const Dev& P, int b, int tid, const double* __restrict__ u, const double* __restrict__ x,
                                   const double* __restrict__ add, double* lu, double* lx) {
    const int row0 = b * P.tile_rows;
    const int hb = P.halo_ptr[b], hn = P.halo_ptr[b + 1] - hb, ns = P.halo_ns[b];
    int idx[STAGE_K];
#pragma unroll
    for (int k = 0; k < STAGE_K; ++k) { const int i = tid + k * BLK; idx[k] = i < hn ? P.halo_rows[hb + i] : -1; }
    for (int i = tid; i < 3 * P.tile_rows; i += BLK) {
        lu[i] = u[3 * (size_t)row0 + i];
        lx[i] = x[3 * (size_t)row0 + i] + (add ? add[3 * (size_t)row0 + i] : 0.0);
    }
    double uu[STAGE_K][3], xx[STAGE_K][3];
#pragma unroll
    for (int k = 0; k < STAGE_K; ++k) {
        if (idx[k] >= 0) {
            const size_t r = (size_t)idx[k];
#pragma unroll
            for (int c = 0; c < 3; ++c) uu[k][c] = u[3 * r + c];
            if (tid + k * BLK < ns) {                              // positions: spring neighbours only
#pragma unroll
                for (int c = 0; c < 3; ++c) xx[k][c] = x[3 * r + c] + (add ? add[3 * r + c] : 0.0);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < STAGE_K; ++k) {
        const int i = tid + k * BLK;
        if (idx[k] >= 0) {
            double* d = lu + 3 * (size_t)(P.tile_rows + i);
            d[0] = uu[k][0]; d[1] = uu[k][1]; d[2] = uu[k][2];
            if (i < ns) {
                double* e = lx + 3 * (size_t)(P.tile_rows + i);
                e[0] = xx[k][0]; e[1] = xx[k][1]; e[2] = xx[k][2];
            }
        }
    }
    for (int i = tid + STAGE_K * BLK; i < hn; i += BLK) {         // very large halos
        const size_t r = (size_t)P.halo_rows[hb + i];
        double* d = lu + 3 * (size_t)(P.tile_rows + i);
#pragma unroll
        for (int k = 0; k < 3; ++k) d[k] = u[3 * r + k];
        if (i < ns) {
            double* e = lx + 3 * (size_t)(P.tile_rows + i);
#pragma unroll
            for (int k = 0; k < 3; ++k) e[k] = x[3 * r + k] + (add ? add[3 * r + k] : 0.0);
        }
    }
}

// reprojection block of one row in factored form (same expressions as k_reproj):
//   a += J_l^T w (J_l u_l + J_p u_p) ; pose partial = J_p^T w J_l u_l ; cross = u_l . H_lp u_p
__device__ inline void row_factored(const RowRec& rc, const Pose& Tcw, const double* xs, const double* ul, const double* up,
                                    double pm, double& a0, double& a1, double& a2, double* part /*9*/) {
    double R[9];
    quat_to_R(Tcw.q, R);
    const double px = R[0] * xs[0] + R[1] * xs[1] + R[2] * xs[2] + Tcw.t[0];
    const double py = R[3] * xs[0] + R[4] * xs[1] + R[5] * xs[2] + Tcw.t[1];
    const double pz = R[6] * xs[0] + R[7] * xs[1] + R[8] * xs[2] + Tcw.t[2];
    double tl[2], tp[2], Jl[2][3], Jp[2][6];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const double j0 = -(double)rc.J[3 * rr], j1 = -(double)rc.J[3 * rr + 1], j2 = -(double)rc.J[3 * rr + 2];
        Jp[rr][0] = pm * (-j1 * pz + j2 * py);
        Jp[rr][1] = pm * (j0 * pz - j2 * px);
        Jp[rr][2] = pm * (-j0 * py + j1 * px);
        Jp[rr][3] = pm * j0; Jp[rr][4] = pm * j1; Jp[rr][5] = pm * j2;
        Jl[rr][0] = j0 * R[0] + j1 * R[3] + j2 * R[6];
        Jl[rr][1] = j0 * R[1] + j1 * R[4] + j2 * R[7];
        Jl[rr][2] = j0 * R[2] + j1 * R[5] + j2 * R[8];
        tl[rr] = Jl[rr][0] * ul[0] + Jl[rr][1] * ul[1] + Jl[rr][2] * ul[2];
        double s = 0;
#pragma unroll
        for (int p = 0; p < 6; ++p) s += Jp[rr][p] * up[p];
        tp[rr] = s;
    }
    const double w = rc.w;
    const double c0 = w * (tl[0] + tp[0]), c1 = w * (tl[1] + tp[1]);
    a0 += Jl[0][0] * c0 + Jl[1][0] * c1;
    a1 += Jl[0][1] * c0 + Jl[1][1] * c1;
    a2 += Jl[0][2] * c0 + Jl[1][2] * c1;
    part[2] = w * (tl[0] * tp[0] + tl[1] * tp[1]);
#pragma unroll
    for (int p = 0; p < 6; ++p) part[3 + p] = w * (Jp[0][p] * tl[0] + Jp[1][p] * tl[1]);
}

__device__ inline double damper_sign(int role) { return (role == 0 || role == 3) ? -1.0 : 1.0; }

// =====================================================================================
// linearisation, part 1: reprojection edges.  One thread per row, one pose per workgroup.
//   ReprojectionError / ReprojectionErrorWithDeformation computeError + linearizeOplus
//   (reference reprojection_error.cc:32-64, reprojection_error_with_deformation.cc:37-68),
//   quadratic form with Huber weight (base_fixed_sized_edge.hpp:49-63, base_edge.h:158-164).
// =====================================================================================
template <bool LIN>
__global__ __launch_bounds__(BLK) void k_reproj(Dev P, const Pose* __restrict__ poses,
                                                const double* __restrict__ xl) {
    __shared__ double lds[4 * 28];
    const int g = xcd_tile(blockIdx.x, P.n_groups);
    if (g >= P.n_groups) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = g * ROW_ALIGN + tid;
    const int kf = P.grp_pose[g];
    const Pose Tcw = poses[kf];
    const bool pfix = P.pose_fixed[kf] != 0;
    double R[9];
    quat_to_R(Tcw.q, R);
    double acc[28];
#pragma unroll
    for (int k = 0; k < 28; ++k) acc[k] = 0;
    const int rf = P.rflag[row];
    const bool rfix = (rf & RF_FIXED) != 0;
    // an edge whose vertices are all fixed is not part of the optimisation (sparse_optimizer.cpp:236)
    const bool active = (rf & RF_OBS) && (rf & RF_REPROJ_ACTIVE) && !(pfix && rfix);
    bool wrote = false;
    if (active) {
        double x0 = xl[3 * row], x1 = xl[3 * row + 1], x2 = xl[3 * row + 2];
        if (P.X0) { x0 += P.X0[3 * row]; x1 += P.X0[3 * row + 1]; x2 += P.X0[3 * row + 2]; }
        const double px = R[0] * x0 + R[1] * x1 + R[2] * x2 + Tcw.t[0];
        const double py = R[3] * x0 + R[4] * x1 + R[5] * x2 + Tcw.t[1];
        const double pz = R[6] * x0 + R[7] * x1 + R[8] * x2 + Tcw.t[2];
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
            const double pm = pfix ? 0.0 : 1.0, lm = rfix ? 0.0 : 1.0;
            double Jp[2][6], Jl[2][3];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const double j0 = -(double)Jf[3 * rr], j1 = -(double)Jf[3 * rr + 1], j2 = -(double)Jf[3 * rr + 2];
                Jp[rr][0] = pm * (-j1 * pz + j2 * py);
                Jp[rr][1] = pm * (j0 * pz - j2 * px);
                Jp[rr][2] = pm * (-j0 * py + j1 * px);
                Jp[rr][3] = pm * j0; Jp[rr][4] = pm * j1; Jp[rr][5] = pm * j2;
                Jl[rr][0] = lm * (j0 * R[0] + j1 * R[3] + j2 * R[6]);
                Jl[rr][1] = lm * (j0 * R[1] + j1 * R[4] + j2 * R[7]);
                Jl[rr][2] = lm * (j0 * R[2] + j1 * R[5] + j2 * R[8]);
            }
            int k = 0;
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int q = p; q < 6; ++q) { acc[k] = w * (Jp[0][p] * Jp[0][q] + Jp[1][p] * Jp[1][q]); ++k; }
#pragma unroll
            for (int p = 0; p < 6; ++p) acc[21 + p] = -w * (Jp[0][p] * r0 + Jp[1][p] * r1);
            if (P.use_lds) {
                // factored form: the PCG kernels rebuild J_l, J_p from these 32 bytes
                RowRec rc;
#pragma unroll
                for (int q = 0; q < 6; ++q) rc.J[q] = Jf[q];
                rc.w = lm * w;
                P.rowrec[row] = rc;
            } else {
                // H_pl (6x3), component-major so that a wave writes 18 contiguous runs
#pragma unroll
                for (int p = 0; p < 6; ++p)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        P.Hpl[(size_t)(p * 3 + c) * P.n_rows + row] = w * (Jp[0][p] * Jl[0][c] + Jp[1][p] * Jl[1][c]);
            }
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
            wrote = true;
        }
    }
    if (LIN && !wrote) {
        if (P.use_lds) {
            RowRec rc;
#pragma unroll
            for (int q = 0; q < 6; ++q) rc.J[q] = 0.f;
            rc.w = 0;
            P.rowrec[row] = rc;
        } else {
#pragma unroll
            for (int c = 0; c < 18; ++c) P.Hpl[(size_t)c * P.n_rows + row] = 0;
        }
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
//   PositionRegularizerWithDeformation (position_regularizer_with_deformation.cc:31-57),
//   SpatialRegularizer (spatial_regularizer.cc:32-59), SpatialRegularizerWithDeformation
//   (spatial_regularizer_with_deformation.cc:36-49), SpatialRegularizerFixed
//   (spatial_regularizer_fixed.cc:32-43).
// =====================================================================================
template <int T, bool LIN, bool LDS>
__global__ __launch_bounds__(BLK) void k_reg(Dev P, const double* __restrict__ xl_g, int cls) {
    __shared__ double lds[4 * 2];
    extern __shared__ double dyn[];
    constexpr int R = 64 / T;
    int b = xcd_tile(blockIdx.x, LDS ? P.n_tiles_cls[cls] : P.n_regblk);
    if (b >= (LDS ? P.n_tiles_cls[cls] : P.n_regblk)) return;
    if (LDS) b = P.tile_list[(cls ? P.n_tiles_cls[0] : 0) + b];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slice = b * 4 + wave;
    const int row = slice * R + lane / T;
    const int t = lane % T;
    const bool rfix = (P.rflag[row] & RF_FIXED) != 0;
    // record headers are double-buffered in two-record chunks; the first chunk of both streams is
    // requested before the tile is staged
    constexpr int U = 2;
    const int s_beg = P.ss_ptr[slice], s_end = P.ss_ptr[slice + 1];
    const int d_beg = P.sd_ptr[slice], d_end = P.sd_ptr[slice + 1];
    uint2 shA[U], shB[U], dhA[U], dhB[U];
    float dwA[U], dwB[U];
    auto load_sh = [&](uint2* h, int idx) {
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int j = idx + 64 * q;
            h[q] = make_uint2(0xFFFFu, 0u);
            if (j < s_end) h[q] = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(P.s_rec + j) + 8);
        }
    };
    auto load_dh = [&](uint2* h, float* w, int idx) {
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int j = idx + 64 * q;
            h[q] = make_uint2(0u, 0xFFFF0000u);
            w[q] = 0.f;
            if (j < d_end) { h[q] = *reinterpret_cast<const uint2*>(P.d_rec + j); w[q] = P.d_w[j]; }
        }
    };
    if (LDS) { load_sh(shA, s_beg + lane); load_dh(dhA, dwA, d_beg + lane); }
    // xl: estimates (dampers act on them), xp: X0 + estimates (springs); with LDS staging both are
    // tile-local arrays indexed by the local neighbour ids
    const double* xl = xl_g;
    const double* xp = nullptr;
    int self = row;
    if (LDS) {
        double* lx = dyn;
        stage_rows(P, b, tid, xl_g, nullptr, lx);
        xl = lx;
        if (P.X0) {
            double* lp = dyn + 3 * (size_t)(P.tile_rows + P.cap_h[cls]);
            stage_rows(P, b, tid, xl_g, P.X0, lp);
            xp = lp;
        }
        __syncthreads();
        self = row - b * P.tile_rows;
    }
    const double xo0 = xl[3 * self], xo1 = xl[3 * self + 1], xo2 = xl[3 * self + 2];
    double xs0 = xo0, xs1 = xo1, xs2 = xo2;                 // spring position = X0 + x
    if (P.X0) {
        if (LDS) { xs0 = xp[3 * self]; xs1 = xp[3 * self + 1]; xs2 = xp[3 * self + 2]; }
        else { xs0 += P.X0[3 * row]; xs1 += P.X0[3 * row + 1]; xs2 += P.X0[3 * row + 2]; }
    }
    double D[6] = {0, 0, 0, 0, 0, 0}, bb[3] = {0, 0, 0}, chi = 0;
    // ---- springs
    const size_t nz = (size_t)P.ss_nnz;
    auto spring = [&](int idx, int o, int meta, double d0) {
        if (!(meta & SM_ACTIVE)) {
            if (LIN) {
                if (LDS) P.s_rec[idx].qc = 0;
                else { P.s_g[idx] = 0; P.s_g[nz + idx] = 0; P.s_g[2 * nz + idx] = 0; }
            }
            return;
        }
        if (!LIN && !(meta & SM_COUNT)) return;           // chi2 only: every edge is counted from one of its rows
        double y0, y1, y2;
        if (LDS && P.X0) { y0 = xp[3 * o]; y1 = xp[3 * o + 1]; y2 = xp[3 * o + 2]; }
        else {
            y0 = xl[3 * o]; y1 = xl[3 * o + 1]; y2 = xl[3 * o + 2];
            if (P.X0) { y0 += P.X0[3 * o]; y1 += P.X0[3 * o + 1]; y2 += P.X0[3 * o + 2]; }
        }
        const double v0 = xs0 - y0, v1 = xs1 - y1, v2 = xs2 - y2;
        const double d = sqrt(v0 * v0 + v1 * v1 + v2 * v2);
        const double r = P.k_spring * (d - d0) / d0;
        double rho0, rho1;
        huber(P.info_pos * r * r, P.delta_pos, rho0, rho1);
        if (meta & SM_COUNT) chi += rho0;
        if (LIN) {
            const double cg = P.spring_form == 0 ? (P.k_spring / d0) * (1.0 / sqrt(d)) * 2.0
                                                 : (P.k_spring / (2 * d0 * d)) * 2.0;
            const double q = rfix ? 0.0 : rho1 * P.info_pos;
            const double g0 = cg * v0, g1 = cg * v1, g2 = cg * v2;
            if (LDS) P.s_rec[idx].qc = q * cg * cg;
            else { const double sq = sqrt(q); P.s_g[idx] = sq * g0; P.s_g[nz + idx] = sq * g1; P.s_g[2 * nz + idx] = sq * g2; }
            D[0] += q * g0 * g0; D[1] += q * g0 * g1; D[2] += q * g0 * g2;
            D[3] += q * g1 * g1; D[4] += q * g1 * g2; D[5] += q * g2 * g2;
            const double qr = q * r;
            bb[0] -= qr * g0; bb[1] -= qr * g1; bb[2] -= qr * g2;
        }
    };
    {
        const int beg = P.ss_ptr[slice], end = P.ss_ptr[slice + 1];
        if (LDS) {
            auto do_sh = [&](const uint2* hdr, int idx) {
#pragma unroll
                for (int q = 0; q < U; ++q) {
                    const int o = (int)(hdr[q].x & 0xFFFFu), m16 = (int)(hdr[q].x >> 16);
                    if (o == REC_NONE) continue;
                    const int meta = ((m16 & SR_ACTIVE) ? SM_ACTIVE : 0) | ((m16 & SR_COUNT) ? SM_COUNT : 0);
                    spring(idx + 64 * q, o, meta, (double)__uint_as_float(hdr[q].y));
                }
            };
            for (int base = beg; base < end; base += 128 * U) {    // wave-uniform trip count
                load_sh(shB, base + 64 * U + lane);
                do_sh(shA, base + lane);
                load_sh(shA, base + 128 * U + lane);
                do_sh(shB, base + 64 * U + lane);
            }
        } else {
            for (int idx = beg + lane; idx < end; idx += 64) {
                const int o = P.s_other[idx];
                if (o < 0) continue;
                spring(idx, o, P.s_meta[idx], (double)P.s_d0[idx]);
            }
        }
    }
    // ---- dampers: r = w((x1n - x1c) - (x2n - x2c)), roles (1c,2c,1n,2n), signs (-,+,+,-)
    auto damper = [&](int idx, int meta, const int* o, double w) {
        if (!(meta & DM_ACTIVE)) {
            if (LIN) { if (LDS) P.d_rec[idx].s = 0; else P.d_s[idx] = 0; }
            return;
        }
        if (!LIN && !(meta & DM_COUNT)) return;           // chi2 only: counted from one of the edge's rows
        const int role = meta & 3;
        const double sgn_own = damper_sign(role);
        double s0 = sgn_own * xo0, s1 = sgn_own * xo1, s2 = sgn_own * xo2;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double sg = damper_sign(k + (k >= role ? 1 : 0));   // role of the k-th other vertex
            if (o[k] >= 0) {
                s0 += sg * xl[3 * o[k]]; s1 += sg * xl[3 * o[k] + 1]; s2 += sg * xl[3 * o[k] + 2];
            }
        }
        const double r0 = w * s0, r1 = w * s1, r2 = w * s2;
        double rho0, rho1;
        huber(P.info_spatial * (r0 * r0 + r1 * r1 + r2 * r2), P.delta_spatial, rho0, rho1);
        if (meta & DM_COUNT) chi += rho0;
        if (LIN) {
            const double fx = rfix ? 0.0 : 1.0;
            const double sfac = fx * rho1 * P.info_spatial * w * w;
            if (LDS) P.d_rec[idx].s = sfac; else P.d_s[idx] = sfac;
            D[0] += sfac; D[3] += sfac; D[5] += sfac;
            const double c = fx * sgn_own * rho1 * P.info_spatial * w;
            bb[0] -= c * r0; bb[1] -= c * r1; bb[2] -= c * r2;
        }
    };
    {
        const int beg = P.sd_ptr[slice], end = P.sd_ptr[slice + 1];
        if (LDS) {
            auto do_dh = [&](const uint2* hdr, const float* ww, int idx) {
#pragma unroll
                for (int q = 0; q < U; ++q) {
                    const int m16 = (int)(hdr[q].y >> 16);
                    if (m16 == REC_NONE) continue;
                    const int r0 = (int)(hdr[q].x & 0xFFFFu), r1 = (int)(hdr[q].x >> 16), r2 = (int)(hdr[q].y & 0xFFFFu);
                    const int o[3] = {r0 == REC_NONE ? -1 : r0, r1 == REC_NONE ? -1 : r1, r2 == REC_NONE ? -1 : r2};
                    damper(idx + 64 * q, m16, o, (double)ww[q]);
                }
            };
            for (int base = beg; base < end; base += 128 * U) {
                load_dh(dhB, dwB, base + 64 * U + lane);
                do_dh(dhA, dwA, base + lane);
                load_dh(dhA, dwA, base + 128 * U + lane);
                do_dh(dhB, dwB, base + 64 * U + lane);
            }
        } else {
            for (int idx = beg + lane; idx < end; idx += 64) {
                const int meta = P.d_meta[idx];
                if (meta < 0) continue;
                const int o[3] = {P.d_o0[idx], P.d_o1[idx], P.d_o2[idx]};
                damper(idx, meta, o, (double)P.d_w[idx]);
            }
        }
    }
    double part[2];
    part[0] = chi;
    part[1] = 0;
    if (LIN) {
#pragma unroll
        for (int k = 0; k < 6; ++k) D[k] = sub_sum_t<T>(D[k]);
#pragma unroll
        for (int k = 0; k < 3; ++k) bb[k] = sub_sum_t<T>(bb[k]);
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
// finalize: fixed-order sums of the partials.  LIN: H_pp, b_p, chi2, max diag.  else chi2, scale.
// =====================================================================================
template <bool LIN>
__global__ __launch_bounds__(BLK) void k_finalize(Dev P) {
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
    if (LIN)
        for (int k = tid; k < P.K; k += BLK) md = fmax(md, P.red[3 + k]);      // k_pose_sums: max |diag H_pp|
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
        // the host reads these after a stream synchronisation: written straight into mapped host memory
        P.h_scal[SC_CHI] = P.scal[SC_CHI];
        if (LIN) P.h_scal[SC_MAXDIAG] = P.scal[SC_MAXDIAG];
        if (!LIN) P.h_scal[SC_SCALE] = P.scal[SC_SCALE];
    }
    if (tid < 8) P.h_flags[tid] = P.flags[tid];
}

// H_pp (21 packed) and b_p (6) of one pose per workgroup: fixed-order sums of the k_reproj partials
// (8 lanes per component, then the 8 in order); red[3 + k] = max |diagonal| for the LM lambda_0
__global__ __launch_bounds__(BLK) void k_pose_sums(Dev P) {
    __shared__ double lds[8][32];
    __shared__ double mdl[32];
    const int k = blockIdx.x, tid = threadIdx.x, c = tid & 31, gl = tid >> 5;
    double s = 0;
    if (c < 27)
        for (int g = P.pose_grp_ptr[k] + gl; g < P.pose_grp_ptr[k + 1]; g += 8) s += P.part_lin[(size_t)g * 32 + c];
    lds[gl][c] = s;
    __syncthreads();
    if (tid < 32) {
        double t = 0;
        if (c < 27) {
#pragma unroll
            for (int q = 0; q < 8; ++q) t += lds[q][c];
            if (c < 21) P.Hpp[k * 21 + c] = t; else P.bp[k * 6 + (c - 21)] = t;
        }
        // diagonal entries of the packed upper triangle: 0,6,11,15,18,20
        mdl[c] = (c == 0 || c == 6 || c == 11 || c == 15 || c == 18 || c == 20) ? fabs(t) : 0.0;
    }
    __syncthreads();
    if (tid == 0) {
        double m = 0;
        for (int q = 0; q < 21; ++q) m = fmax(m, mdl[q]);
        P.red[3 + k] = m;
    }
}

__global__ void k_publish(Dev P) {
    if (threadIdx.x < 8) P.h_flags[threadIdx.x] = P.flags[threadIdx.x];
}

// =====================================================================================
// Two-level preconditioner of the fused path (one pose, <= CO_MAX coarse unknowns).
//   k_coarse_tile   (per linearisation, one workgroup per tile): the tile's rows of Z^T H Z, i.e.
//                   sum of H_ij over i in the tile and j in each row group its incidences reach,
//                   its part of the landmark-pose coupling, sum of b, number of free rows;
//   k_coarse_reduce (one workgroup): C0 = Z^T H Z, N = Z^T Z, bc = Z^T b in fixed summation order;
//   k_coarse_invert (per trial, one workgroup): (C0 + lambda N)^-1 by in-place Gauss-Jordan in LDS.
// Coarse unknown 3g+c = translation c of every free row of group g; 3G+a = pose component a.
// =====================================================================================
template <int T>
__global__ __launch_bounds__(BLK) void k_coarse_tile(Dev P) {
    __shared__ double lds[4 * 9];
    extern __shared__ double dyn[];
    constexpr int R = 64 / T;
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slice = b * 4 + wave;
    const int row = slice * R + lane / T;
    const int t = lane % T;
    const int row0 = b * P.tile_rows;
    const int own_grp = row0 / ROW_ALIGN;
    const int hb = P.halo_ptr[b], hn = P.halo_ptr[b + 1] - hb;
    double* lx = dyn;                                                      // positions, tile + halo
    unsigned char* lfix = reinterpret_cast<unsigned char*>(dyn + 3 * (size_t)(P.tile_rows + P.max_halo));
    unsigned short* lgrp = reinterpret_cast<unsigned short*>(lfix + P.tile_rows + P.max_halo + 8);
    stage_rows(P, b, tid, P.lin_xl, P.X0, lx);
    for (int i = tid; i < P.tile_rows + hn; i += BLK) {
        const int r = i < P.tile_rows ? row0 + i : P.halo_rows[hb + i - P.tile_rows];
        lfix[i] = (P.rflag[r] & RF_FIXED) ? 1 : 0;
        lgrp[i] = (unsigned short)(r / ROW_ALIGN);
    }
    __syncthreads();
    const int self = row - row0;
    const bool rfix = lfix[self] != 0;
    const double xs[3] = {lx[3 * self], lx[3 * self + 1], lx[3 * self + 2]};
    const int sbeg = P.ss_ptr[slice], send = rfix ? sbeg : P.ss_ptr[slice + 1];
    const int dbeg = P.sd_ptr[slice], dend = rfix ? dbeg : P.sd_ptr[slice + 1];
    // which groups do this tile's incidences reach?  (bit mask, order-independent OR)
    __shared__ unsigned int reach;
    if (tid == 0) reach = 1u << own_grp;
    __syncthreads();
    {
        unsigned int m = 0;
        for (int idx = sbeg + lane; idx < send; idx += 64) {
            const SpringRec rc = P.s_rec[idx];
            if (rc.other != REC_NONE && !lfix[rc.other]) m |= 1u << lgrp[rc.other];
        }
        for (int idx = dbeg + lane; idx < dend; idx += 64) {
            const DamperRec rc = P.d_rec[idx];
            if (rc.meta == REC_NONE || (rc.meta & DM_UNARY)) continue;
            const uint16_t o[3] = {rc.o0, rc.o1, rc.o2};
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (o[k] != REC_NONE && !lfix[o[k]]) m |= 1u << lgrp[o[k]];
        }
        if (m) atomicOr(&reach, m);
    }
    __syncthreads();
    const unsigned int reached = reach;
    for (int hg = 0; hg < P.n_groups; ++hg) {
        double acc[6] = {0, 0, 0, 0, 0, 0};
        if (!((reached >> hg) & 1u)) {                                 // uniform: nothing to sum
            if (tid < 6) P.co_ct[((size_t)b * P.n_groups + hg) * 6 + tid] = 0;
            continue;
        }
        {
            if (t == 0 && !rfix && hg == own_grp) {
#pragma unroll
                for (int k = 0; k < 6; ++k) acc[k] = P.D[6 * (size_t)row + k];
            }
            for (int idx = sbeg + lane; idx < send; idx += 64) {
                const SpringRec rc = P.s_rec[idx];
                if (rc.other == REC_NONE || lfix[rc.other] || lgrp[rc.other] != hg) continue;
                const int o = rc.other;
                const double v0 = xs[0] - lx[3 * o], v1 = xs[1] - lx[3 * o + 1], v2 = xs[2] - lx[3 * o + 2];
                const double m = -rc.qc;                                   // H_ij = -qc v v^T
                acc[0] += m * v0 * v0; acc[1] += m * v0 * v1; acc[2] += m * v0 * v2;
                acc[3] += m * v1 * v1; acc[4] += m * v1 * v2; acc[5] += m * v2 * v2;
            }
            for (int idx = dbeg + lane; idx < dend; idx += 64) {
                const DamperRec rc = P.d_rec[idx];
                if (rc.meta == REC_NONE || (rc.meta & DM_UNARY)) continue;
                const int role = rc.meta & 3;
                const uint16_t o[3] = {rc.o0, rc.o1, rc.o2};
                const double so = damper_sign(role);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if (o[k] == REC_NONE || lfix[o[k]] || lgrp[o[k]] != hg) continue;
                    const double c = so * damper_sign(k + (k >= role ? 1 : 0)) * rc.s;   // H_ik = sg_i sg_k s I
                    acc[0] += c; acc[3] += c; acc[5] += c;
                }
            }
        }
        block_sum_store<6>(acc, lds, tid, P.co_ct + ((size_t)b * P.n_groups + hg) * 6);
        __syncthreads();
    }
    {   // tile-level block: the same sum restricted to j inside the tile (second, finer level)
        double acc[6] = {0, 0, 0, 0, 0, 0};
        if (t == 0 && !rfix) {
#pragma unroll
            for (int k = 0; k < 6; ++k) acc[k] = P.D[6 * (size_t)row + k];
        }
        for (int idx = sbeg + lane; idx < send; idx += 64) {
            const SpringRec rc = P.s_rec[idx];
            if (rc.other == REC_NONE || rc.other >= P.tile_rows || lfix[rc.other]) continue;
            const int o = rc.other;
            const double v0 = xs[0] - lx[3 * o], v1 = xs[1] - lx[3 * o + 1], v2 = xs[2] - lx[3 * o + 2];
            const double m = -rc.qc;
            acc[0] += m * v0 * v0; acc[1] += m * v0 * v1; acc[2] += m * v0 * v2;
            acc[3] += m * v1 * v1; acc[4] += m * v1 * v2; acc[5] += m * v2 * v2;
        }
        for (int idx = dbeg + lane; idx < dend; idx += 64) {
            const DamperRec rc = P.d_rec[idx];
            if (rc.meta == REC_NONE || (rc.meta & DM_UNARY)) continue;
            const int role = rc.meta & 3;
            const uint16_t o[3] = {rc.o0, rc.o1, rc.o2};
            const double so = damper_sign(role);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (o[k] == REC_NONE || o[k] >= P.tile_rows || lfix[o[k]]) continue;
                const double c = so * damper_sign(k + (k >= role ? 1 : 0)) * rc.s;
                acc[0] += c; acc[3] += c; acc[5] += c;
            }
        }
        block_sum_store<6>(acc, lds, tid, P.co_bt + (size_t)b * 6);
        __syncthreads();
    }
    // landmark-pose coupling of the tile's rows: sum of H_lp = J_l^T w J_p (3x6), two halves of 9
    double cp[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) cp[k] = 0;
    double tb[4] = {0, 0, 0, 0};
    if (t == 0 && !rfix) {
        const RowRec rc = P.rowrec[row];
        if (rc.w != 0.0 && !P.pose_fixed[0]) {
            const Pose Tcw = P.lin_pose[0];
            double Rm[9];
            quat_to_R(Tcw.q, Rm);
            const double px = Rm[0] * xs[0] + Rm[1] * xs[1] + Rm[2] * xs[2] + Tcw.t[0];
            const double py = Rm[3] * xs[0] + Rm[4] * xs[1] + Rm[5] * xs[2] + Tcw.t[1];
            const double pz = Rm[6] * xs[0] + Rm[7] * xs[1] + Rm[8] * xs[2] + Tcw.t[2];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const double j0 = -(double)rc.J[3 * rr], j1 = -(double)rc.J[3 * rr + 1], j2 = -(double)rc.J[3 * rr + 2];
                const double Jp[6] = {-j1 * pz + j2 * py, j0 * pz - j2 * px, -j0 * py + j1 * px, j0, j1, j2};
                const double Jl[3] = {j0 * Rm[0] + j1 * Rm[3] + j2 * Rm[6], j0 * Rm[1] + j1 * Rm[4] + j2 * Rm[7],
                                      j0 * Rm[2] + j1 * Rm[5] + j2 * Rm[8]};
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int a = 0; a < 6; ++a) cp[c * 6 + a] += rc.w * Jl[c] * Jp[a];
            }
        }
        tb[0] = P.bl[3 * row]; tb[1] = P.bl[3 * row + 1]; tb[2] = P.bl[3 * row + 2];
        tb[3] = 1.0;
    }
    block_sum_store<9>(cp, lds, tid, P.co_cp + (size_t)b * 18);
    __syncthreads();
    block_sum_store<9>(cp + 9, lds, tid, P.co_cp + (size_t)b * 18 + 9);
    __syncthreads();
    block_sum_store<4>(tb, lds, tid, P.co_tb + (size_t)b * 4);
}

__global__ __launch_bounds__(BLK) void k_coarse_reduce(Dev P) {
    const int tid = threadIdx.x;
    const int G = P.n_groups, n = P.co_n, rb = ROW_ALIGN / P.tile_rows;
    for (int i = tid; i < n * n; i += BLK) P.co_c0[i] = 0;
    __syncthreads();
    // group-group blocks: thread per (g, h), fixed order over the group's tiles and their slots
    for (int gh = tid; gh < G * G; gh += BLK) {
        const int g = gh / G, h = gh % G;
        double acc[6] = {0, 0, 0, 0, 0, 0};
        for (int tl = g * rb; tl < (g + 1) * rb; ++tl)
#pragma unroll
            for (int k = 0; k < 6; ++k) acc[k] += P.co_ct[((size_t)tl * G + h) * 6 + k];
        const double m[9] = {acc[0], acc[1], acc[2], acc[1], acc[3], acc[4], acc[2], acc[4], acc[5]};
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) P.co_c0[(size_t)(3 * g + a) * n + 3 * h + c] = m[a * 3 + c];
    }
    // group sums of the coupling, of b and of the free-row counts
    for (int g = tid; g < G; g += BLK) {
        double cp[18], tb[4] = {0, 0, 0, 0};
        for (int k = 0; k < 18; ++k) cp[k] = 0;
        for (int tl = g * rb; tl < (g + 1) * rb; ++tl) {
            for (int k = 0; k < 18; ++k) cp[k] += P.co_cp[(size_t)tl * 18 + k];
            for (int k = 0; k < 4; ++k) tb[k] += P.co_tb[(size_t)tl * 4 + k];
        }
        for (int c = 0; c < 3; ++c) {
            for (int a = 0; a < 6; ++a) {
                P.co_c0[(size_t)(3 * g + c) * n + 3 * G + a] = cp[c * 6 + a];
                P.co_c0[(size_t)(3 * G + a) * n + 3 * g + c] = cp[c * 6 + a];
            }
            P.co_nn[3 * g + c] = tb[3];
            P.co_bc[3 * g + c] = tb[c];
        }
    }
    // pose block
    if (tid < 36) {
        const int a = tid / 6, c = tid % 6;
        const int lo = a < c ? a : c, hi = a < c ? c : a;
        const int pk = lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo);
        const bool pfix = P.pose_fixed[0] != 0;
        P.co_c0[(size_t)(3 * G + a) * n + 3 * G + c] = pfix ? (a == c ? 1.0 : 0.0) : P.Hpp[pk];
        if (c == 0) { P.co_nn[3 * G + a] = pfix ? 0.0 : 1.0; P.co_bc[3 * G + a] = pfix ? 0.0 : P.bp[a]; }
    }
}

__global__ __launch_bounds__(BLK) void k_coarse_invert(Dev P, double lam) {
    // In-place Gauss-Jordan (SPD: no pivoting) with the matrix in REGISTERS: thread (br, bc) keeps a
    // 6x6 block; per pivot step only the pivot row and column go through LDS (double-buffered: one
    // barrier per step).  The matrix is padded to a multiple of 6 with an identity block.  A
    // non-positive pivot switches the coarse level off for this trial.
    extern __shared__ double A[];                                          // n x n (result, for y0)
    constexpr int BS = 6, NBMAX = (CO_MAX + BS - 1) / BS;
    __shared__ double colb[2][NBMAX * BS], rowb[2][NBMAX * BS];
    __shared__ int bad;
    const int tid = threadIdx.x, n = P.co_n;
    const int nb = (n + BS - 1) / BS;
    const bool act = tid < nb * nb;
    const int br = act ? tid / nb : 0, bc = act ? tid % nb : 0;
    if (tid == 0) bad = 0;
    for (int tl = tid; tl < P.n_regblk; tl += BLK) {                       // tile-level 3x3 blocks of this trial
        double Bi[6];
        const double nf = P.co_tb[4 * (size_t)tl + 3];
        const bool okb = nf > 0 && inv3_sym(P.co_bt + 6 * (size_t)tl, lam * nf, Bi) && Bi[0] > 0;
#pragma unroll
        for (int q = 0; q < 6; ++q) P.co_bti[6 * (size_t)tl + q] = okb ? Bi[q] : 0.0;
    }
    double a[BS][BS];
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) {
            const int r = br * BS + i, c = bc * BS + j;
            double v = r == c ? 1.0 : 0.0;
            if (act && r < n && c < n) {
                v = P.co_c0[(size_t)r * n + c];
                if (r == c) {
                    v += lam * P.co_nn[r];
                    if (P.co_nn[r] == 0.0 && v == 0.0) v = 1.0;            // empty group: keep the system regular
                }
            }
            a[i][j] = v;
        }
    __syncthreads();
    const int np = nb * BS;
    bool ok = true;
    for (int p = 0; p < np; ++p) {
        const int pb = p / BS, pi = p % BS, buf = p & 1;
        if (act && bc == pb) {
#pragma unroll
            for (int i = 0; i < BS; ++i) {
                double v = a[i][0];
#pragma unroll
                for (int j = 1; j < BS; ++j) v = (pi == j) ? a[i][j] : v;
                colb[buf][br * BS + i] = v;
            }
        }
        if (act && br == pb) {
#pragma unroll
            for (int j = 0; j < BS; ++j) {
                double v = a[0][j];
#pragma unroll
                for (int i = 1; i < BS; ++i) v = (pi == i) ? a[i][j] : v;
                rowb[buf][bc * BS + j] = v;
            }
        }
        __syncthreads();
        const double piv = rowb[buf][p];
        if (!(piv > 0) || !isfinite(piv)) { ok = false; break; }
        const double pinv = 1.0 / piv;
        double cr[BS], rw[BS];
#pragma unroll
        for (int i = 0; i < BS; ++i) { cr[i] = colb[buf][br * BS + i]; rw[i] = rowb[buf][bc * BS + i]; }
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
            for (int j = 0; j < BS; ++j) {
                const bool rp = br * BS + i == p, cp = bc * BS + j == p;
                const double upd = a[i][j] - cr[i] * rw[j] * pinv;
                a[i][j] = rp ? (cp ? pinv : rw[j] * pinv) : (cp ? -cr[i] * pinv : upd);
            }
    }
    if (!ok && tid == 0) bad = 1;
    __syncthreads();
    const bool off = bad != 0;
    if (act) {
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
            for (int j = 0; j < BS; ++j) {
                const int r = br * BS + i, c = bc * BS + j;
                if (r < n && c < n) { const double v = off ? 0.0 : a[i][j]; A[r * n + c] = v; P.co_inv[(size_t)r * n + c] = v; }
            }
    }
    __syncthreads();
    for (int j = tid; j < n; j += BLK) {
        double y = 0;
        for (int c = 0; c < n; ++c) y += A[j * n + c] * P.co_bc[c];
        P.co_y0[j] = y;
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

__global__ __launch_bounds__(BLK) void k_trial_setup(Dev P, double lam) {
    const int i = blockIdx.x * BLK + threadIdx.x;
    if (i < P.n_rows) {
        double Di[6];
        const bool ok = inv3_sym(P.D + 6 * (size_t)i, lam, Di);
        if (!ok || !isfinite(Di[0])) P.flags[2] = 1;
#pragma unroll
        for (int k = 0; k < 6; ++k) P.Dinv[6 * (size_t)i + k] = Di[k];
        const double r0 = P.bl[3 * i], r1 = P.bl[3 * i + 1], r2 = P.bl[3 * i + 2];
        P.rv[3 * i] = r0; P.rv[3 * i + 1] = r1; P.rv[3 * i + 2] = r2;
        double y0 = 0, y1 = 0, y2 = 0;
        if (P.coarse && !(P.rflag[i] & RF_FIXED)) {
            const int g = i / ROW_ALIGN, tl = i / P.tile_rows;
            double yt[3];
            tile_level(P.co_bti + 6 * (size_t)tl, P.co_tb + 4 * (size_t)tl, yt);
            y0 = P.co_y0[3 * g] + yt[0]; y1 = P.co_y0[3 * g + 1] + yt[1]; y2 = P.co_y0[3 * g + 2] + yt[2];
        }
        P.uv3[3 * i] = Di[0] * r0 + Di[1] * r1 + Di[2] * r2 + y0;
        P.uv3[3 * i + 1] = Di[1] * r0 + Di[3] * r1 + Di[4] * r2 + y1;
        P.uv3[3 * i + 2] = Di[2] * r0 + Di[4] * r1 + Di[5] * r2 + y2;
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
            if (P.coarse && !P.pose_fixed[i]) s += P.co_y0[3 * P.n_groups + a];
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
template <int T, bool LDS>
__global__ __launch_bounds__(BLK) void k_spmv(Dev P, double lam, int it) {
    static_assert(!LDS, "gather fallback only: the LDS-staged path is k_spmv_f");
    __shared__ double lds[4 * 9];
    constexpr int R = 64 / T;
    const int b = xcd_tile(blockIdx.x, P.n_regblk);
    if (b >= P.n_regblk) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slice = b * 4 + wave;
    const int row = slice * R + lane / T;
    const int t = lane % T;
    const bool rfix = (P.rflag[row] & RF_FIXED) != 0;
    const double* u = P.uv3;
    const int self = row;
    double a0 = 0, a1 = 0, a2 = 0;
    double part[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const double ul0 = u[3 * self], ul1 = u[3 * self + 1], ul2 = u[3 * self + 2];
    {
        // row part: the T lanes of a row share the 6 pose components of H_pl
        const int kf = P.grp_pose[row / ROW_ALIGN];
        if (t == 0) {
            const double* D = P.D + 6 * (size_t)row;
            a0 = (D[0] + lam) * ul0 + D[1] * ul1 + D[2] * ul2;
            a1 = D[1] * ul0 + (D[3] + lam) * ul1 + D[4] * ul2;
            a2 = D[2] * ul0 + D[4] * ul1 + (D[5] + lam) * ul2;
        }
        double h0 = 0, h1 = 0, h2 = 0;
#pragma unroll
        for (int p = t; p < 6; p += T) {
            const double upk = ((it & 1) ? P.up2 : P.up)[6 * kf + p];
            const double e0 = P.Hpl[(size_t)(p * 3) * P.n_rows + row];
            const double e1 = P.Hpl[(size_t)(p * 3 + 1) * P.n_rows + row];
            const double e2 = P.Hpl[(size_t)(p * 3 + 2) * P.n_rows + row];
            h0 += e0 * upk; h1 += e1 * upk; h2 += e2 * upk;
            part[3 + p] = e0 * ul0 + e1 * ul1 + e2 * ul2;
        }
        a0 += h0; a1 += h1; a2 += h2;
        part[2] = ul0 * h0 + ul1 * h1 + ul2 * h2;
    }
    if (!rfix) {
        const int beg = P.ss_ptr[slice], end = P.ss_ptr[slice + 1];
        const size_t nz = (size_t)P.ss_nnz;
        for (int idx = beg + lane; idx < end; idx += 64) {
            const int o = P.s_other[idx];
            if (o < 0) continue;
            const double g0 = P.s_g[idx], g1 = P.s_g[nz + idx], g2 = P.s_g[2 * nz + idx];
            const double dot = g0 * u[3 * o] + g1 * u[3 * o + 1] + g2 * u[3 * o + 2];
            a0 -= g0 * dot; a1 -= g1 * dot; a2 -= g2 * dot;
        }
    }
    if (!rfix) {
        const int beg = P.sd_ptr[slice], end = P.sd_ptr[slice + 1];
        for (int idx = beg + lane; idx < end; idx += 64) {
            const int meta = P.d_meta[idx];
            if (meta < 0 || (meta & DM_UNARY)) continue;
            const int o[3] = {P.d_o0[idx], P.d_o1[idx], P.d_o2[idx]};
            const int role = meta & 3;
            double s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double sg = damper_sign(k + (k >= role ? 1 : 0));
                if (o[k] >= 0) { s0 += sg * u[3 * o[k]]; s1 += sg * u[3 * o[k] + 1]; s2 += sg * u[3 * o[k] + 2]; }
            }
            const double c = damper_sign(role) * P.d_s[idx];
            a0 += c * s0; a1 += c * s1; a2 += c * s2;
        }
    }
    a0 = sub_sum_t<T>(a0); a1 = sub_sum_t<T>(a1); a2 = sub_sum_t<T>(a2);
    if (t == 0) {
        P.wv[3 * row] = a0; P.wv[3 * row + 1] = a1; P.wv[3 * row + 2] = a2;
        part[0] = P.rv[3 * row] * ul0 + P.rv[3 * row + 1] * ul1 + P.rv[3 * row + 2] * ul2;
        part[1] = a0 * ul0 + a1 * ul1 + a2 * ul2;
    }
    block_sum_store<9>(part, lds, tid, P.part_spmv + (size_t)b * NPART);
}

// =====================================================================================
// PCG kernel 1, LDS-staged path: the same operator in factored form.  The tile's u and the positions
// of the linearisation point are staged (own rows + halo); per incidence the kernel reads ONE
// 16-byte record: spring  a_i += qc (v . (u_i - u_j)) v,  v = x_i - x_j;
//                 damper  a_i += sg_i s (sum_k sg_k u_k)  (all four vertices, the own one included);
// per row 32 bytes of reprojection factors instead of the 6x3 H_pl block and the 3x3 diagonal.
// =====================================================================================
template <int T>
__global__ __launch_bounds__(BLK, 4) void k_spmv_f(Dev P, double lam, int cls, int it) {
    __shared__ double lds[4 * 9];
    extern __shared__ double dyn[];
    constexpr int R = 64 / T;
    constexpr int U = 2;                                           // records per lane and buffer (two buffers per stream)
    const int bi = xcd_tile(blockIdx.x, P.n_tiles_cls[cls]);
    if (bi >= P.n_tiles_cls[cls]) return;
    const int b = P.tile_list[(cls ? P.n_tiles_cls[0] : 0) + bi];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slice = b * 4 + wave;
    const int row = slice * R + lane / T;
    const int t = lane % T;
    const bool rfix = (P.rflag[row] & RF_FIXED) != 0;
    const int kf = P.grp_pose[row / ROW_ALIGN];
    const int sbeg = P.ss_ptr[slice], send = rfix ? sbeg : P.ss_ptr[slice + 1];
    const int dbeg = P.sd_ptr[slice], dend = rfix ? dbeg : P.sd_ptr[slice + 1];
    double* lu = dyn;
    double* lx = dyn + 3 * (size_t)(P.tile_rows + P.cap_h[cls] + 1);
    // row ZROW of both arrays is zero: padding records and absent damper vertices point at it, so the
    // incidence loops are branch-free and the LDS reads of a whole chunk can be in flight together
    const int ZROW = P.tile_rows + P.cap_h[cls], ZROWX = P.tile_rows + P.cap_s[cls];
    if (tid < 3) { lu[3 * ZROW + tid] = 0; lx[3 * ZROWX + tid] = 0; }
    stage_rows2(P, b, tid, P.uv3, P.lin_xl, P.X0, lu, lx);
    // row factors and the first record chunks are requested while the staging loads are in flight
    RowRec rc;
    rc.w = 0;
    double rv0 = 0, rv1 = 0, rv2 = 0;
    if (t == 0) {
        rc = P.rowrec[row];
        rv0 = P.rv[3 * row]; rv1 = P.rv[3 * row + 1]; rv2 = P.rv[3 * row + 2];
    }
    // records are double-buffered: chunk k+1 is requested before chunk k is consumed (with ~3 waves
    // per SIMD the loops are bound by the latency of their own loads otherwise)
    SpringRec srA[U], srB[U];
    DamperRec drA[U], drB[U];
    auto load_springs = [&](SpringRec* sr, int idx) {
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int j = idx + 64 * q;
            sr[q].other = REC_NONE; sr[q].qc = 0;
            if (j < send) sr[q] = P.s_rec[j];
        }
    };
    auto load_dampers = [&](DamperRec* dr, int idx) {
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int j = idx + 64 * q;
            dr[q].meta = 0; dr[q].s = 0; dr[q].o0 = dr[q].o1 = dr[q].o2 = REC_NONE;
            if (j < dend) dr[q] = P.d_rec[j];
        }
    };
    load_springs(srA, sbeg + lane);
    load_dampers(drA, dbeg + lane);
    __syncthreads();
    const int self = row - b * P.tile_rows;
    const double ul[3] = {lu[3 * self], lu[3 * self + 1], lu[3 * self + 2]};
    const double xs[3] = {lx[3 * self], lx[3 * self + 1], lx[3 * self + 2]};
    double a0 = 0, a1 = 0, a2 = 0;
    auto do_springs = [&](const SpringRec* sr) {
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int o = sr[q].other == REC_NONE ? ZROW : (int)sr[q].other;     // padding: qc = 0
            const int ox = sr[q].other == REC_NONE ? ZROWX : (int)sr[q].other;
            const double v0 = xs[0] - lx[3 * ox], v1 = xs[1] - lx[3 * ox + 1], v2 = xs[2] - lx[3 * ox + 2];
            const double dot = sr[q].qc * (v0 * (ul[0] - lu[3 * o]) + v1 * (ul[1] - lu[3 * o + 1]) + v2 * (ul[2] - lu[3 * o + 2]));
            a0 += dot * v0; a1 += dot * v1; a2 += dot * v2;
        }
    };
    auto do_dampers = [&](const DamperRec* dr) {
#pragma unroll
        for (int q = 0; q < U; ++q) {
            // padding records carry s = 0; a unary damper (the other vertex is a value) and absent
            // vertices read the zero row, which leaves the diagonal term s u_i
            const int meta = dr[q].meta == REC_NONE ? 0 : (int)dr[q].meta;
            const double sv = dr[q].meta == REC_NONE ? 0.0 : dr[q].s;
            const bool un = (meta & DM_UNARY) != 0;
            const int role = meta & 3;
            const int o0 = (un || dr[q].o0 == REC_NONE) ? ZROW : (int)dr[q].o0;
            const int o1 = (un || dr[q].o1 == REC_NONE) ? ZROW : (int)dr[q].o1;
            const int o2 = (un || dr[q].o2 == REC_NONE) ? ZROW : (int)dr[q].o2;
            const double so = damper_sign(role);
            const double g0 = damper_sign(role == 0 ? 1 : 0), g1 = damper_sign(role <= 1 ? 2 : 1), g2 = damper_sign(role <= 2 ? 3 : 2);
            const double s0 = so * ul[0] + g0 * lu[3 * o0] + g1 * lu[3 * o1] + g2 * lu[3 * o2];
            const double s1 = so * ul[1] + g0 * lu[3 * o0 + 1] + g1 * lu[3 * o1 + 1] + g2 * lu[3 * o2 + 1];
            const double s2 = so * ul[2] + g0 * lu[3 * o0 + 2] + g1 * lu[3 * o1 + 2] + g2 * lu[3 * o2 + 2];
            const double c = so * sv;
            a0 += c * s0; a1 += c * s1; a2 += c * s2;
        }
    };
    for (int base = sbeg; base < send; base += 128 * U) {          // wave-uniform trip count
        load_springs(srB, base + 64 * U + lane);
        do_springs(srA);
        load_springs(srA, base + 128 * U + lane);
        do_springs(srB);
    }
    for (int base = dbeg; base < dend; base += 128 * U) {
        load_dampers(drB, base + 64 * U + lane);
        do_dampers(drA);
        load_dampers(drA, base + 128 * U + lane);
        do_dampers(drB);
    }
    // the row's own terms come last: their temporaries then never coexist with the record registers
    double part[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (t == 0) {
        a0 += lam * ul[0]; a1 += lam * ul[1]; a2 += lam * ul[2];
        if (rc.w != 0.0) {
            double up[6];
#pragma unroll
            for (int p = 0; p < 6; ++p) up[p] = ((it & 1) ? P.up2 : P.up)[6 * kf + p];
            row_factored(rc, P.lin_pose[kf], xs, ul, up, P.pose_fixed[kf] ? 0.0 : 1.0, a0, a1, a2, part);
        }
    }
    a0 = sub_sum_t<T>(a0); a1 = sub_sum_t<T>(a1); a2 = sub_sum_t<T>(a2);
    if (t == 0) {
        P.wv[3 * row] = a0; P.wv[3 * row + 1] = a1; P.wv[3 * row + 2] = a2;
        part[0] = rv0 * ul[0] + rv1 * ul[1] + rv2 * ul[2];
        part[1] = a0 * ul[0] + a1 * ul[1] + a2 * ul[2];
    }
    block_sum_store<9>(part, lds, tid, P.part_spmv + (size_t)b * NPART);
}

// =====================================================================================
// large problems only: fixed-order reduction of the SpMV partials.  Workgroup 0: the three dot
// partials over all workgroups; workgroup 1 + k: the six pose sums of pose k.
// =====================================================================================
__global__ __launch_bounds__(BLK) void k_reduce_partials(Dev P) {
    __shared__ double lds[4 * 6];
    if (P.flags[0]) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blockIdx.x == 0) {
        double v[3] = {0, 0, 0};
        for (int b = tid; b < P.n_regblk; b += BLK) {
            v[0] += P.part_spmv[(size_t)b * NPART];
            v[1] += P.part_spmv[(size_t)b * NPART + 1];
            v[2] += P.part_spmv[(size_t)b * NPART + 2];
        }
        block_sum<3>(v, lds, lane, wave);
        if (tid == 0) { P.red[0] = v[0]; P.red[1] = v[1]; P.red[2] = v[2]; }
    } else {
        const int k = blockIdx.x - 1;
        const int rb = ROW_ALIGN / (BLK / P.T);
        const int g0 = P.pose_grp_ptr[k] * rb, g1 = P.pose_grp_ptr[k + 1] * rb;
        double acc[6] = {0, 0, 0, 0, 0, 0};
        for (int g = g0 + tid; g < g1; g += BLK) {
#pragma unroll
            for (int a = 0; a < 6; ++a) acc[a] += P.part_spmv[(size_t)g * NPART + 3 + a];
        }
        block_sum<6>(acc, lds, lane, wave);
        if (tid == 0) {
#pragma unroll
            for (int a = 0; a < 6; ++a) P.red[3 + 6 * k + a] = acc[a];
        }
    }
}

// =====================================================================================
// PCG kernel 2 (Chronopoulos-Gear single-reduction CG): every workgroup re-derives the scalars
// from the partials in a fixed order, then updates its rows:
//   gamma = r.u, delta = w.u, beta = gamma/gamma_old, alpha = gamma/(delta - beta*gamma/alpha_old)
//   p = u + beta p ; s = w + beta s ; x += alpha p ; r -= alpha s ; u = M^-1 r
// Workgroups >= n_vec8 own the pose rows (w_p = (H_pp + lambda) u_p + sum_l H_pl u_l).
// =====================================================================================
__global__ __launch_bounds__(BLK) void k_pcg_update(Dev P, double lam, int it, double tol2, double peek_tol2) {
    __shared__ double lds[4 * 3];
    const int n_vecblk = P.n_vecblk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // u_p and r_p are read by every workgroup (for the scalars) and rewritten by the pose workgroups
    // of the same launch: they are a ping-pong pair (read half it&1, write the other)
    const double* up_in = (it & 1) ? P.up2 : P.up;   double* up_out = (it & 1) ? P.up : P.up2;
    const double* rp_in = (it & 1) ? P.rp2 : P.rp;   double* rp_out = (it & 1) ? P.rp : P.rp2;
    // Everything this launch reads is requested before the first dependent use (flag, scalars,
    // partials, the two rows of this thread): otherwise the launch is a chain of four round trips.
    const int done_flag = P.flags[0];
    const double sc_gamma0 = P.scal[SC_GAMMA0];
    const double sc_slot0 = P.scal[(it & 1) ? SC_SLOT1 : SC_SLOT0], sc_slot1 = P.scal[((it & 1) ? SC_SLOT1 : SC_SLOT0) + 1];
    const int n_vec2 = (n_vecblk + 1) >> 1;
    const int n_vec8 = ((n_vec2 + 7) >> 3) << 3;
    const bool row_wg = (int)blockIdx.x < n_vec8;
    const int pair = row_wg ? xcd_tile(blockIdx.x, n_vec2) * BLK + tid : 0;
    const bool has_rows = row_wg && 2 * pair < P.n_rows;
    const size_t o = 6 * (size_t)pair;
    double uu[6], pp[6], ww[6], ss[6], rr[6], xx[6], Di[12];
    if (has_rows) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double2 a = *reinterpret_cast<const double2*>(P.uv3 + o + 2 * k);
            const double2 b = *reinterpret_cast<const double2*>(P.pv + o + 2 * k);
            const double2 c = *reinterpret_cast<const double2*>(P.wv + o + 2 * k);
            const double2 d = *reinterpret_cast<const double2*>(P.sv + o + 2 * k);
            const double2 e = *reinterpret_cast<const double2*>(P.rv + o + 2 * k);
            const double2 f = *reinterpret_cast<const double2*>(P.xv + o + 2 * k);
            uu[2 * k] = a.x; uu[2 * k + 1] = a.y; pp[2 * k] = b.x; pp[2 * k + 1] = b.y;
            ww[2 * k] = c.x; ww[2 * k + 1] = c.y; ss[2 * k] = d.x; ss[2 * k + 1] = d.y;
            rr[2 * k] = e.x; rr[2 * k + 1] = e.y; xx[2 * k] = f.x; xx[2 * k + 1] = f.y;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double2 a = *reinterpret_cast<const double2*>(P.Dinv + 2 * o + 2 * k);
            Di[2 * k] = a.x; Di[2 * k + 1] = a.y;
        }
    }
    // pose workgroups (one wave per pose): their inputs are requested up front as well
    const int pk_pose = row_wg ? P.K : (int)(blockIdx.x - n_vec8) * 4 + wave;
    const bool has_pose = pk_pose < P.K;
    const int pa = lane < 6 ? lane : 0;
    double q_up[6], q_H[6], q_Hi[6], q_pp = 0, q_sp = 0, q_rp = 0, q_xp = 0, q_acc[6] = {0, 0, 0, 0, 0, 0};
    int pg0 = 0, pg1 = 0;
    if (has_pose) {
        const int rb = ROW_ALIGN / (BLK / P.T);       // reg-blocks per row group
        pg0 = P.pose_grp_ptr[pk_pose] * rb; pg1 = P.pose_grp_ptr[pk_pose + 1] * rb;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int lo = pa < c ? pa : c, hi = pa < c ? c : pa;
            const int pk = lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo);
            q_up[c] = up_in[6 * pk_pose + c];
            q_H[c] = P.Hpp[21 * pk_pose + pk];
            q_Hi[c] = P.Hppinv[36 * pk_pose + pa * 6 + c];
        }
        q_pp = P.pp[6 * pk_pose + pa]; q_sp = P.sp[6 * pk_pose + pa]; q_rp = rp_in[6 * pk_pose + pa]; q_xp = P.xp[6 * pk_pose + pa];
        if (P.hier) {
#pragma unroll
            for (int a = 0; a < 6; ++a) q_acc[a] = P.red[3 + 6 * pk_pose + a];
        } else if (pg0 + lane < pg1) {
#pragma unroll
            for (int a = 0; a < 6; ++a) q_acc[a] += P.part_spmv[(size_t)(pg0 + lane) * NPART + 3 + a];
        }
    }
    double v[3] = {0, 0, 0};
    if (P.hier) {
        if (tid == 0) { v[0] = P.red[0]; v[1] = P.red[1]; v[2] = P.red[2]; }
    } else {
        for (int b = tid; b < P.n_regblk; b += BLK) {
            v[0] += P.part_spmv[(size_t)b * NPART];
            v[1] += P.part_spmv[(size_t)b * NPART + 1];
            v[2] += P.part_spmv[(size_t)b * NPART + 2];
        }
    }
    // pose rows: gamma_p = r_p.u_p ; delta_p = u_p.(H_pp + lam)u_p + cross (cross is v[2])
    for (int i = tid; i < 6 * P.K; i += BLK) {
        const int k = i / 6, a = i % 6;
        const double ua = up_in[i];
        v[0] += rp_in[i] * ua;
        double s = lam * ua;
        for (int c = 0; c < 6; ++c) {
            const int lo = a < c ? a : c, hi = a < c ? c : a;
            const int pk = lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo);     // packed upper index
            s += P.Hpp[21 * k + pk] * up_in[6 * k + c];
        }
        v[1] += ua * s;
    }
    if (done_flag) return;
    block_sum<3>(v, lds, lane, wave);
    const double gamma = v[0], delta = v[1] + v[2];
    double* nslot = P.scal + ((it & 1) ? SC_SLOT0 : SC_SLOT1);
    const double gamma0 = it == 0 ? gamma : sc_gamma0;
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
    const double beta = it == 0 ? 0.0 : gamma / sc_slot0;
    const double alpha = it == 0 ? gamma / delta : gamma / (delta - beta * gamma / sc_slot1);
    if (blockIdx.x == 0 && tid == 0) {
        nslot[0] = gamma;
        nslot[1] = alpha;
        if (it == 0) P.scal[SC_GAMMA0] = gamma;
        P.flags[1] = it + 1;
        // "peek" milestones for early trial rejection: level 1 at peek_tol, level 2 at peek_tol/10
        if (gamma <= peek_tol2 * gamma0) {
            P.flags[3] = max(P.flags[3], (gamma <= 1e-6 * peek_tol2 * gamma0 ? 4 : gamma <= 1e-4 * peek_tol2 * gamma0 ? 3 : gamma <= 1e-2 * peek_tol2 * gamma0 ? 2 : 1));
            if (P.flags[4] == 0) P.flags[4] = it + 1;               // iterations the first milestone took (sizes the next first batch)
        }
    }
    // row workgroups: every thread updates TWO consecutive rows (6 doubles = three 16-byte
    // accesses per vector); n_rows is a multiple of 256, so pairs never straddle anything
    if (row_wg) {
        if (has_rows) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                pp[k] = uu[k] + beta * pp[k];
                ss[k] = ww[k] + beta * ss[k];
                xx[k] += alpha * pp[k];
                rr[k] -= alpha * ss[k];
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const double* Dh = Di + 6 * h;
                const double r0 = rr[3 * h], r1 = rr[3 * h + 1], r2 = rr[3 * h + 2];
                uu[3 * h] = Dh[0] * r0 + Dh[1] * r1 + Dh[2] * r2;
                uu[3 * h + 1] = Dh[1] * r0 + Dh[3] * r1 + Dh[4] * r2;
                uu[3 * h + 2] = Dh[2] * r0 + Dh[4] * r1 + Dh[5] * r2;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                *reinterpret_cast<double2*>(P.pv + o + 2 * k) = make_double2(pp[2 * k], pp[2 * k + 1]);
                *reinterpret_cast<double2*>(P.sv + o + 2 * k) = make_double2(ss[2 * k], ss[2 * k + 1]);
                *reinterpret_cast<double2*>(P.xv + o + 2 * k) = make_double2(xx[2 * k], xx[2 * k + 1]);
                *reinterpret_cast<double2*>(P.rv + o + 2 * k) = make_double2(rr[2 * k], rr[2 * k + 1]);
                *reinterpret_cast<double2*>(P.uv3 + o + 2 * k) = make_double2(uu[2 * k], uu[2 * k + 1]);
            }
        }
    } else {
        // pose workgroups: one wave per pose; its 64 lanes split the pose's SpMV partials
        if (has_pose) {
            const int k = pk_pose;
            double acc[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) acc[a] = q_acc[a];
            if (!P.hier) {
                for (int g = pg0 + lane + 64; g < pg1; g += 64) {
#pragma unroll
                    for (int a = 0; a < 6; ++a) acc[a] += P.part_spmv[(size_t)g * NPART + 3 + a];
                }
#pragma unroll
                for (int a = 0; a < 6; ++a) acc[a] = wave_sum(acc[a]);
            }
            // lanes 0..5 own one pose component each
            const int a = pa;
            double hw = acc[0], ua = q_up[0];
#pragma unroll
            for (int q = 1; q < 6; ++q) { hw = (a == q) ? acc[q] : hw; ua = (a == q) ? q_up[q] : ua; }
            const int i = 6 * k + a;
            double w = lam * ua + hw;
#pragma unroll
            for (int c = 0; c < 6; ++c) w += q_H[c] * q_up[c];
            const double p = ua + beta * q_pp;
            const double sN = w + beta * q_sp;
            const double rnew = q_rp - alpha * sN;
            double unew = 0;
#pragma unroll
            for (int c = 0; c < 6; ++c) unew += q_Hi[c] * __shfl(rnew, c, 64);
            if (lane < 6) {
                P.pp[i] = p;
                P.sp[i] = sN;
                P.xp[i] = q_xp + alpha * p;
                rp_out[i] = rnew;
                up_out[i] = unew;
            }
        }
    }
}

// =====================================================================================
// Fused PCG iteration for small problems (single-frame tracking, short BA windows): ONE launch per
// iteration.  F(it) = [vector update of iteration it-1] followed by [operator apply of iteration
// it].  Every workgroup re-derives the scalars from the previous launch's partials, updates its own
// rows, and RECOMPUTES the updated u of its halo rows from (r, s, w, M^-1) instead of waiting for
// their owners -- so there is no inter-workgroup hand-off inside a launch.  r, s, w, the pose
// vectors and the partials are ping-pong pairs (read half (it+1)&1, write half it&1): owners write
// the new values while neighbours still read the old ones.  Same arithmetic, in the same order, as
// k_pcg_update + k_spmv.
// =====================================================================================
template <int T, bool CO>
__global__ __launch_bounds__(BLK) void k_pcg_fused(Dev P, double lam, int it, double tol2, double peek_tol2) {
    __shared__ double lds[4 * 9];
    __shared__ double s_up[6];
    extern __shared__ double dyn[];
    constexpr int R = 64 / T;
    constexpr int U = 4;
    const int b = xcd_tile(blockIdx.x, P.n_regblk);
    if (b >= P.n_regblk) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // read half / write half of the ping-pong pairs.  Launch 0 only applies the operator: the state
    // written by k_trial_setup lives in half 0 and stays there.
    const int hin = it == 0 ? 0 : ((it + 1) & 1), hout = it & 1;
    const double* r_in = hin ? P.rv2 : P.rv;    double* r_out = hout ? P.rv2 : P.rv;
    const double* s_in = hin ? P.sv2 : P.sv;    double* s_out = hout ? P.sv2 : P.sv;
    const double* w_in = hin ? P.wv2 : P.wv;    double* w_out = hout ? P.wv2 : P.wv;
    const double* rp_in = hin ? P.rp2 : P.rp;   double* rp_out = hout ? P.rp2 : P.rp;
    const double* sp_in = hin ? P.sp2 : P.sp;   double* sp_out = hout ? P.sp2 : P.sp;
    const double* up_in = hin ? P.up2 : P.up;   double* up_out = hout ? P.up2 : P.up;
    const double* part_in = hin ? P.part_spmv2 : P.part_spmv;
    double* part_out = hout ? P.part_spmv2 : P.part_spmv;
    const double* ts_in = hin ? P.part_ts2 : P.part_ts;
    double* ts_out = hout ? P.part_ts2 : P.part_ts;
    const int slice = b * 4 + wave;
    const int row = slice * R + lane / T;
    const int t = lane % T;
    const int row0 = b * P.tile_rows;
    // The launch is a chain of dependent memory round trips unless everything is requested at once:
    // level 1 = whatever is addressed by the tile index alone (tile descriptor, flags, scalars, all
    // partials, own rows, fixed-stride halo list, slice pointers), level 2 = what those address
    // (records, halo rows, the pose's partials and vectors).  Nothing is loaded after that.
    const int4 td = *reinterpret_cast<const int4*>(P.tile_desc + 8 * (size_t)b);
    const int kf = td.x, pg0 = td.y, pg1 = td.z, hb = td.w;       // a tile never straddles two poses
    const int hn = P.tile_desc[8 * (size_t)b + 4];
    const int done_flag = P.flags[0];
    const int ipq = it > 0 ? it - 1 : 0;                           // PCG iteration whose scalars this launch finishes
    const double sc_gamma0 = P.scal[SC_GAMMA0];
    const double sc_slot0 = P.scal[(ipq & 1) ? SC_SLOT1 : SC_SLOT0], sc_slot1 = P.scal[((ipq & 1) ? SC_SLOT1 : SC_SLOT0) + 1];
    const int self = row - row0;
    const bool rfix = (P.rflag[row] & RF_FIXED) != 0;
    double* lu = dyn;
    double* lx = dyn + 3 * (size_t)(P.tile_rows + P.max_halo);     // positions of the linearisation point

    // ================= phase 1: every global load this launch needs is requested up front (the
    // launch is a chain of dependent round trips otherwise: partials -> vectors -> records)
    double v[3] = {0, 0, 0};
    const bool coarse = CO && it > 0;                              // CO: two-level preconditioner compiled in
    double ts9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};                  // thread q: sums of r, s, w of tile q (coarse level)
    if (it > 0) {
        for (int q = tid; q < P.n_regblk; q += BLK) {
            v[0] += part_in[(size_t)q * NPART];
            v[1] += part_in[(size_t)q * NPART + 1];
            v[2] += part_in[(size_t)q * NPART + 2];
        }
        if (coarse && tid < P.n_regblk) {
#pragma unroll
            for (int c = 0; c < 9; ++c) ts9[c] = ts_in[(size_t)c * P.n_regblk + tid];   // component-major: coalesced
        }
        for (int i = tid; i < 6 * P.K; i += BLK) {
            const int k = i / 6, a = i % 6;
            const double ua = up_in[i];
            v[0] += rp_in[i] * ua;
            double sacc = lam * ua;
            for (int c = 0; c < 6; ++c) {
                const int lo = a < c ? a : c, hi = a < c ? c : a;
                const int pk = lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo);
                sacc += P.Hpp[21 * k + pk] * up_in[6 * k + c];
            }
            v[1] += ua * sacc;
        }
    }
    // own row (one thread per row) and first halo row of this thread
    const bool own = tid < P.tile_rows;
    const size_t orow = (size_t)(row0 + (own ? tid : 0));
    double o_u[3], o_p[3], o_w[3], o_s[3], o_r[3], o_x[3], o_D[6];
    const bool hh = tid < hn;
    const size_t hrow = (size_t)P.halo_fix[(size_t)b * BLK + tid];
    double h_w[3], h_s[3], h_r[3], h_D[6];
    double x_own[3] = {0, 0, 0}, x_h[3] = {0, 0, 0};
    const bool o_free = own && !(P.rflag[orow] & RF_FIXED);
    const bool h_free = hh && !(P.rflag[hrow] & RF_FIXED);
    const int cn = P.co_n, cG = P.n_groups;
    double bt_own[6], bt_h[6];
    const int th_h = (int)(hrow / (size_t)P.tile_rows);              // tile of this thread's halo row
    if (coarse) {
#pragma unroll
        for (int q = 0; q < 6; ++q) { bt_own[q] = P.co_bti[6 * (size_t)b + q]; bt_h[q] = P.co_bti[6 * (size_t)th_h + q]; }
    }
    if (own) {
#pragma unroll
        for (int k = 0; k < 3; ++k) x_own[k] = P.lin_xl[3 * orow + k] + (P.X0 ? P.X0[3 * orow + k] : 0.0);
    }
    if (hh) {
#pragma unroll
        for (int k = 0; k < 3; ++k) x_h[k] = P.lin_xl[3 * hrow + k] + (P.X0 ? P.X0[3 * hrow + k] : 0.0);
    }
    if (it > 0) {
        if (own) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                o_u[k] = P.uv3[3 * orow + k]; o_p[k] = P.pv[3 * orow + k]; o_w[k] = w_in[3 * orow + k];
                o_s[k] = s_in[3 * orow + k]; o_r[k] = r_in[3 * orow + k]; o_x[k] = P.xv[3 * orow + k];
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) o_D[k] = P.Dinv[6 * orow + k];
        }
        if (hh) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { h_w[k] = w_in[3 * hrow + k]; h_s[k] = s_in[3 * hrow + k]; h_r[k] = r_in[3 * hrow + k]; }
#pragma unroll
            for (int k = 0; k < 6; ++k) h_D[k] = P.Dinv[6 * hrow + k];
        }
    } else {
        stage_rows(P, b, tid, P.uv3, nullptr, lu);
    }
    // wave 0: everything the pose-vector update of this tile's pose needs
    const int pa = lane < 6 ? lane : 0;
    double q_up[6], q_H[6], q_Hi[6], q_sp = 0, q_rp = 0, q_pp = 0, q_xp = 0, q_acc[6] = {0, 0, 0, 0, 0, 0};
    if (wave == 0) {
#pragma unroll
        for (int c = 0; c < 6; ++c) q_up[c] = up_in[6 * kf + c];
        if (it > 0) {
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const int lo = pa < c ? pa : c, hi = pa < c ? c : pa;
                const int pk = lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo);
                q_H[c] = P.Hpp[21 * kf + pk];
                q_Hi[c] = P.Hppinv[36 * kf + pa * 6 + c];
            }
            q_sp = sp_in[6 * kf + pa];
            q_rp = rp_in[6 * kf + pa];
            if (b == pg0) { q_pp = P.pp[6 * kf + pa]; q_xp = P.xp[6 * kf + pa]; }
            if (pg0 + lane < pg1) {
#pragma unroll
                for (int c = 0; c < 6; ++c) q_acc[c] += part_in[(size_t)(pg0 + lane) * NPART + 3 + c];
            }
        }
    }
    // row factors, the tile's pose at the linearisation point, first record chunks
    RowRec rc;
    rc.w = 0;
    if (t == 0) rc = P.rowrec[row];
    const Pose Tlin = P.lin_pose[kf];
    const double pmask = P.pose_fixed[kf] ? 0.0 : 1.0;
    const int sbeg = P.ss_ptr[slice], send = rfix ? sbeg : P.ss_ptr[slice + 1];
    const int dbeg = P.sd_ptr[slice], dend = rfix ? dbeg : P.sd_ptr[slice + 1];
    SpringRec sr[U];
    DamperRec dr[U];
    auto load_springs = [&](int idx) {
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int j = idx + 64 * q;
            sr[q].other = REC_NONE;
            if (j < send) sr[q] = P.s_rec[j];
        }
    };
    auto load_dampers = [&](int idx) {
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int j = idx + 64 * q;
            dr[q].meta = REC_NONE;
            if (j < dend) dr[q] = P.d_rec[j];
        }
    };
    load_springs(sbeg + lane);
    load_dampers(dbeg + lane);

    if (done_flag) return;
    // wave 0: w_p = (H_pp + lambda) u_p + sum_l H_pl u_l of the tile's pose (independent of alpha, beta)
    double w_pose = 0, ua_pose = 0;
    if (wave == 0) {
        ua_pose = q_up[0];
#pragma unroll
        for (int q = 1; q < 6; ++q) ua_pose = (pa == q) ? q_up[q] : ua_pose;
        if (it > 0) {
            double acc[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) acc[c] = q_acc[c];
            for (int g = pg0 + lane + 64; g < pg1; g += 64) {
#pragma unroll
                for (int c = 0; c < 6; ++c) acc[c] += part_in[(size_t)g * NPART + 3 + c];
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) acc[c] = wave_sum(acc[c]);
            double hw = acc[0];
#pragma unroll
            for (int q = 1; q < 6; ++q) hw = (pa == q) ? acc[q] : hw;
            w_pose = lam * ua_pose + hw;
#pragma unroll
            for (int c = 0; c < 6; ++c) w_pose += q_H[c] * q_up[c];
        }
    }
    // ================= coarse level: y = A_c^-1 Z^T r_new with r_new = r - alpha w - alpha beta s, i.e.
    // y = yR - alpha yW - alpha beta yS; the three products are formed before alpha, beta are known
    double* c_ts = dyn + 6 * (size_t)(P.tile_rows + P.max_halo);  // n_regblk x 9 tile sums
    double* c_v = c_ts + 9 * (size_t)P.n_regblk;                   // 10 vectors of CO_MAX: Rc Sc Wc yR yS yW y + second halves of yR yS yW
    if (coarse) {
        if (tid < P.n_regblk) {
#pragma unroll
            for (int c = 0; c < 9; ++c) c_ts[9 * tid + c] = ts9[c];
        }
        if (wave == 0 && lane < 6) {
            c_v[0 * CO_MAX + 3 * cG + lane] = q_rp;
            c_v[1 * CO_MAX + 3 * cG + lane] = q_sp;
            c_v[2 * CO_MAX + 3 * cG + lane] = w_pose;
        }
        __syncthreads();
        const int rb = ROW_ALIGN / P.tile_rows;
        if (tid < 9 * cG) {                                        // group sums in fixed order
            const int g = tid / 9, c = tid % 9;
            double sum = 0;
            for (int j = 0; j < rb; ++j) sum += c_ts[9 * (g * rb + j) + c];
            c_v[(c / 3) * CO_MAX + 3 * g + c % 3] = sum;
        }
        __syncthreads();
        if (tid < 2 * cn) {                                        // thread (row, half of the columns): three partial dot products
            const int r = tid % cn, half = tid / cn;
            const int c0 = half ? cn / 2 : 0, c1 = half ? cn : cn / 2;
            double yr = 0, ys = 0, yw = 0;
#pragma unroll 4
            for (int c = c0; c < c1; ++c) {
                const double m = P.co_inv[(size_t)c * cn + r];         // symmetric: column read, coalesced over r
                yr += m * c_v[c]; ys += m * c_v[CO_MAX + c]; yw += m * c_v[2 * CO_MAX + c];
            }
            double* dst = c_v + (half ? 7 : 3) * CO_MAX;             // second halves go to scratch vectors 7..9
            dst[r] = yr; dst[CO_MAX + r] = ys; dst[2 * CO_MAX + r] = yw;
        }
    }
    // ================= phase 2: scalars of iteration it-1 (k_pcg_update prologue)
    double alpha = 0, beta = 0;
    if (it > 0) {
        block_sum<3>(v, lds, lane, wave);
        const double gamma = v[0], delta = v[1] + v[2];
        const int ip = it - 1;                                     // PCG iteration these scalars belong to
        double* nslot = P.scal + ((ip & 1) ? SC_SLOT0 : SC_SLOT1);
        const double gamma0 = ip == 0 ? gamma : sc_gamma0;
        const bool bad = !isfinite(gamma) || !isfinite(delta);
        const bool conv = (gamma <= tol2 * gamma0) || bad || gamma == 0.0;
        if (conv) {
            if (blockIdx.x == 0 && tid == 0) {
                if (bad) P.flags[2] = 1;
                P.flags[1] = ip;
                __threadfence();
                P.flags[0] = 1;
            }
            return;
        }
        beta = ip == 0 ? 0.0 : gamma / sc_slot0;
        alpha = ip == 0 ? gamma / delta : gamma / (delta - beta * gamma / sc_slot1);
        if (blockIdx.x == 0 && tid == 0) {
            nslot[0] = gamma;
            nslot[1] = alpha;
            if (ip == 0) P.scal[SC_GAMMA0] = gamma;
            P.flags[1] = ip + 1;
            if (gamma <= peek_tol2 * gamma0) {
                P.flags[3] = max(P.flags[3], (gamma <= 1e-6 * peek_tol2 * gamma0 ? 4 : gamma <= 1e-4 * peek_tol2 * gamma0 ? 3 : gamma <= 1e-2 * peek_tol2 * gamma0 ? 2 : 1));
                if (P.flags[4] == 0) P.flags[4] = ip + 1;
            }
        }
    }
    if (coarse) {                                                  // (the reduction above was a barrier: yR, yS, yW are visible)
        if (tid < cn) c_v[6 * CO_MAX + tid] = (c_v[3 * CO_MAX + tid] + c_v[7 * CO_MAX + tid]) - alpha * (c_v[5 * CO_MAX + tid] + c_v[9 * CO_MAX + tid]) - alpha * beta * (c_v[4 * CO_MAX + tid] + c_v[8 * CO_MAX + tid]);
        __syncthreads();
    }
    const double* ycor = c_v + 6 * CO_MAX;
    // ================= phase 3: pose vector of this tile's pose (wave 0; every workgroup recomputes
    // it, the first workgroup of the pose also stores the pose part of the state)
    if (wave == 0) {
        const int a = pa;
        const int i = 6 * kf + a;
        const double ua = ua_pose;
        double unew = ua;
        if (it > 0) {
            const double w = w_pose;
            const double sN = w + beta * q_sp;
            const double rnew = q_rp - alpha * sN;
            unew = 0;
#pragma unroll
            for (int c = 0; c < 6; ++c) unew += q_Hi[c] * __shfl(rnew, c, 64);
            if (coarse && pmask != 0.0) unew += ycor[3 * cG + a];
            if (lane < 6 && b == pg0) {
                const double p = ua + beta * q_pp;
                P.pp[i] = p;
                P.xp[i] = q_xp + alpha * p;
                sp_out[i] = sN;
                rp_out[i] = rnew;
                up_out[i] = unew;
            }
        }
        if (lane < 6) s_up[lane] = unew;
    }
    // ================= phase 4: u of the tile (own rows: full update, stored; halo rows:
    // recomputed from r, s, w, M^-1, LDS only)
    double dot_ru = 0;                                             // r.u of this thread's own row (after the update)
    double sum9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};                  // this tile's sums of r, s, w for the next launch
    if (it > 0) {
        if (own) {
            double rn[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double p = o_u[k] + beta * o_p[k];
                const double sN = o_w[k] + beta * o_s[k];
                P.pv[3 * orow + k] = p;
                s_out[3 * orow + k] = sN;
                P.xv[3 * orow + k] = o_x[k] + alpha * p;
                rn[k] = o_r[k] - alpha * sN;
                r_out[3 * orow + k] = rn[k];
                sum9[k] = rn[k]; sum9[3 + k] = sN;
            }
            double u0 = o_D[0] * rn[0] + o_D[1] * rn[1] + o_D[2] * rn[2];
            double u1 = o_D[1] * rn[0] + o_D[3] * rn[1] + o_D[4] * rn[2];
            double u2 = o_D[2] * rn[0] + o_D[4] * rn[1] + o_D[5] * rn[2];
            if (coarse && o_free) {                                // group level + tile level
                double rc3[3], yt[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) rc3[k] = c_ts[9 * b + k] - alpha * c_ts[9 * b + 6 + k] - alpha * beta * c_ts[9 * b + 3 + k];
                tile_level(bt_own, rc3, yt);
                const int g = row0 / ROW_ALIGN;
                u0 += ycor[3 * g] + yt[0]; u1 += ycor[3 * g + 1] + yt[1]; u2 += ycor[3 * g + 2] + yt[2];
            }
            P.uv3[3 * orow] = u0; P.uv3[3 * orow + 1] = u1; P.uv3[3 * orow + 2] = u2;
            lu[3 * tid] = u0; lu[3 * tid + 1] = u1; lu[3 * tid + 2] = u2;
            dot_ru = rn[0] * u0 + rn[1] * u1 + rn[2] * u2;
        }
        for (int i = tid; i < hn; i += BLK) {
            if (i != tid) {                                        // beyond the prefetched one (large halos only)
                const size_t r2 = (size_t)P.halo_rows[hb + i];
#pragma unroll
                for (int k = 0; k < 3; ++k) { h_w[k] = w_in[3 * r2 + k]; h_s[k] = s_in[3 * r2 + k]; h_r[k] = r_in[3 * r2 + k]; }
#pragma unroll
                for (int k = 0; k < 6; ++k) h_D[k] = P.Dinv[6 * r2 + k];
            }
            double rn[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) rn[k] = h_r[k] - alpha * (h_w[k] + beta * h_s[k]);
            double* dst = lu + 3 * (size_t)(P.tile_rows + i);
            double y0 = 0, y1 = 0, y2 = 0;
            if (coarse) {
                const size_t r2 = i == tid ? hrow : (size_t)P.halo_rows[hb + i];
                const bool fr = i == tid ? h_free : !(P.rflag[r2] & RF_FIXED);
                if (fr) {
                    const int sl = (int)(r2 / ROW_ALIGN);
                    const int th = (int)(r2 / (size_t)P.tile_rows);
                    double rc3[3], yt[3], btl[6];
#pragma unroll
                    for (int k = 0; k < 3; ++k) rc3[k] = c_ts[9 * th + k] - alpha * c_ts[9 * th + 6 + k] - alpha * beta * c_ts[9 * th + 3 + k];
#pragma unroll
                    for (int k = 0; k < 6; ++k) btl[k] = i == tid ? bt_h[k] : P.co_bti[6 * (size_t)th + k];
                    tile_level(btl, rc3, yt);
                    y0 = ycor[3 * sl] + yt[0]; y1 = ycor[3 * sl + 1] + yt[1]; y2 = ycor[3 * sl + 2] + yt[2];
                }
            }
            dst[0] = h_D[0] * rn[0] + h_D[1] * rn[1] + h_D[2] * rn[2] + y0;
            dst[1] = h_D[1] * rn[0] + h_D[3] * rn[1] + h_D[4] * rn[2] + y1;
            dst[2] = h_D[2] * rn[0] + h_D[4] * rn[1] + h_D[5] * rn[2] + y2;
        }
    }
    if (own) { lx[3 * tid] = x_own[0]; lx[3 * tid + 1] = x_own[1]; lx[3 * tid + 2] = x_own[2]; }
    for (int i = tid; i < hn; i += BLK) {
        if (i != tid) {
            const size_t r2 = (size_t)P.halo_rows[hb + i];
#pragma unroll
            for (int k = 0; k < 3; ++k) x_h[k] = P.lin_xl[3 * r2 + k] + (P.X0 ? P.X0[3 * r2 + k] : 0.0);
        }
        double* dst = lx + 3 * (size_t)(P.tile_rows + i);
        dst[0] = x_h[0]; dst[1] = x_h[1]; dst[2] = x_h[2];
    }
    __syncthreads();
    // ================= phase 5: operator apply on the staged u (k_spmv_f)
    double a0 = 0, a1 = 0, a2 = 0;
    double part[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const double ul[3] = {lu[3 * self], lu[3 * self + 1], lu[3 * self + 2]};
    const double xs[3] = {lx[3 * self], lx[3 * self + 1], lx[3 * self + 2]};
    if (t == 0) {
        a0 = lam * ul[0]; a1 = lam * ul[1]; a2 = lam * ul[2];
        if (rc.w != 0.0) {
            double up[6];
#pragma unroll
            for (int p = 0; p < 6; ++p) up[p] = s_up[p];
            row_factored(rc, Tlin, xs, ul, up, pmask, a0, a1, a2, part);
        }
    }
    for (int idx = sbeg + lane; idx < send; idx += 64 * U) {
        if (idx != sbeg + lane) load_springs(idx);
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int o = sr[q].other;
            if (o == REC_NONE) continue;
            const double v0 = xs[0] - lx[3 * o], v1 = xs[1] - lx[3 * o + 1], v2 = xs[2] - lx[3 * o + 2];
            const double dot = sr[q].qc * (v0 * (ul[0] - lu[3 * o]) + v1 * (ul[1] - lu[3 * o + 1]) + v2 * (ul[2] - lu[3 * o + 2]));
            a0 += dot * v0; a1 += dot * v1; a2 += dot * v2;
        }
    }
    for (int idx = dbeg + lane; idx < dend; idx += 64 * U) {
        if (idx != dbeg + lane) load_dampers(idx);
#pragma unroll
        for (int q = 0; q < U; ++q) {
            if (dr[q].meta == REC_NONE) continue;
            if (dr[q].meta & DM_UNARY) {
                a0 += dr[q].s * ul[0]; a1 += dr[q].s * ul[1]; a2 += dr[q].s * ul[2];
                continue;
            }
            const int role = dr[q].meta & 3;
            const uint16_t o[3] = {dr[q].o0, dr[q].o1, dr[q].o2};
            const double so = damper_sign(role);
            double s0 = so * ul[0], s1 = so * ul[1], s2 = so * ul[2];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double sg = damper_sign(k + (k >= role ? 1 : 0));
                if (o[k] != REC_NONE) { s0 += sg * lu[3 * o[k]]; s1 += sg * lu[3 * o[k] + 1]; s2 += sg * lu[3 * o[k] + 2]; }
            }
            const double c = so * dr[q].s;
            a0 += c * s0; a1 += c * s1; a2 += c * s2;
        }
    }
    a0 = sub_sum_t<T>(a0); a1 = sub_sum_t<T>(a1); a2 = sub_sum_t<T>(a2);
    if (t == 0) {
        w_out[3 * row] = a0; w_out[3 * row + 1] = a1; w_out[3 * row + 2] = a2;
        if (it == 0) {
            const double r0 = r_out[3 * row], r1 = r_out[3 * row + 1], r2 = r_out[3 * row + 2];
            part[0] = r0 * ul[0] + r1 * ul[1] + r2 * ul[2];
            sum9[0] = r0; sum9[1] = r1; sum9[2] = r2;            // launch 0: r = b, s = 0
        }
        part[1] = a0 * ul[0] + a1 * ul[1] + a2 * ul[2];
        sum9[6] = a0; sum9[7] = a1; sum9[8] = a2;
    }
    part[0] += dot_ru;
    block_sum_store<9>(part, lds, tid, part_out + (size_t)b * NPART);
    if (CO) {
        __syncthreads();
        block_sum<9>(sum9, lds, lane, wave);
        if (tid < 9) {
            double sv = sum9[0];
#pragma unroll
            for (int q = 1; q < 9; ++q) sv = (tid == q) ? sum9[q] : sv;
            ts_out[(size_t)tid * P.n_regblk + b] = sv;
        }
    }
}

// =====================================================================================
// trial state = state (+) x ;  partial of computeScale: sum_j x_j (lambda x_j + b_j)
// (levenberg.cpp:167-174; LandmarkVertex::oplusImpl landmark_vertex.cc:40-43)
// =====================================================================================
__global__ __launch_bounds__(BLK) void k_apply(Dev P, double lam, const Pose* __restrict__ pose_in,
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
        Pose Tcw = pose_in[i];
        if (!P.pose_fixed[i]) {
            double upd[6];
            for (int a = 0; a < 6; ++a) {
                upd[a] = P.xp[6 * i + a];
                sc[0] += upd[a] * (lam * upd[a] + P.bp[6 * i + a]);
            }
            pose_oplus(Tcw, upd);
        }
        pose_out[i] = Tcw;
    }
    block_sum<1>(sc, lds, lane, wave);
    if (tid == 0) P.part_apply[blockIdx.x] = sc[0];
}

// =====================================================================================
// edge taps (edge-parallel, not on the timed path): residuals of every edge at a given state
// =====================================================================================
__global__ void k_tap_residuals(Dev P, const Pose* poses, const double* xl, const int* vrow,
                                const int* sp_ij, const float* sp_d0, const int* dm_idx, const float* dm_w,
                                double* r_reproj, double* r_spring, double* r_damper) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P.M) {
        const int row = vrow[i];
        r_reproj[2 * i] = r_reproj[2 * i + 1] = 0;
        if (P.rflag[row] & RF_OBS) {
            const Pose Tcw = poses[P.grp_pose[row / ROW_ALIGN]];
            double R[9];
            quat_to_R(Tcw.q, R);
            double x0 = xl[3 * row], x1 = xl[3 * row + 1], x2 = xl[3 * row + 2];
            if (P.X0) { x0 += P.X0[3 * row]; x1 += P.X0[3 * row + 1]; x2 += P.X0[3 * row + 2]; }
            const double px = R[0] * x0 + R[1] * x1 + R[2] * x2 + Tcw.t[0];
            const double py = R[3] * x0 + R[4] * x1 + R[5] * x2 + Tcw.t[1];
            const double pz = R[6] * x0 + R[7] * x1 + R[8] * x2 + Tcw.t[2];
            float u, v;
            project_f32(P.cam, (float)px, (float)py, (float)pz, u, v);
            r_reproj[2 * i] = (double)P.uv[2 * row] - (double)u;
            r_reproj[2 * i + 1] = (double)P.uv[2 * row + 1] - (double)v;
        }
    }
    if (i < P.n_sp) {
        const int a = vrow[sp_ij[2 * i]], b = vrow[sp_ij[2 * i + 1]];
        double v[3];
        for (int k = 0; k < 3; ++k) {
            v[k] = xl[3 * a + k] - xl[3 * b + k];
            if (P.X0) v[k] = (xl[3 * a + k] + P.X0[3 * a + k]) - (xl[3 * b + k] + P.X0[3 * b + k]);
        }
        const double d = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), d0 = (double)sp_d0[i];
        r_spring[i] = P.k_spring * (d - d0) / d0;
    }
    if (i < P.n_dm) {
        const double w = (double)dm_w[i];
        for (int k = 0; k < 3; ++k) {
            double s = 0;
            for (int role = 0; role < 4; ++role) {
                const int v = dm_idx[4 * i + role];
                if (v >= 0) s += damper_sign(role) * xl[3 * vrow[v] + k];
            }
            r_damper[3 * i + k] = w * s;
        }
    }
}

// =====================================================================================
// host side
// =====================================================================================
int engine_num_poses(const Engine* e) { return e->d.K; }

void arena_release(Arena* a) {
    if (a->base) (void)hipFree(a->base);
    a->base = nullptr;
    a->cap = a->off = 0;
}

struct ArenaPlan {                   // two passes: size, then carve
    Arena* a;
    bool dry;
    size_t off = 0;
    template <class Tp> Tp* get(size_t n) {
        const size_t bytes = ((n * sizeof(Tp) + 255) / 256) * 256 + 256;
        Tp* p = dry ? nullptr : reinterpret_cast<Tp*>(a->base + off);
        off += bytes;
        return p;
    }
};

static void carve(ArenaPlan& A, Dev& d, bool has_X0, size_t nnz_s, size_t nnz_d, size_t n_slices, size_t n_halo, Engine* e) {
    const size_t nr = (size_t)d.n_rows, K = (size_t)d.K;
    d.grp_pose = A.get<int>(d.n_groups);
    d.pose_grp_ptr = A.get<int>(K + 1);
    d.rflag = A.get<uint8_t>(nr);
    d.pose_fixed = A.get<uint8_t>(K);
    d.uv = A.get<float>(2 * nr);
    double* X0 = A.get<double>(has_X0 ? 3 * nr : 1);
    d.X0 = has_X0 ? X0 : nullptr;
    d.ss_ptr = A.get<int>(n_slices + 1);
    d.sd_ptr = A.get<int>(n_slices + 1);
    d.halo_ptr = A.get<int>((size_t)d.n_regblk + 1);
    d.halo_rows = A.get<int>(n_halo);
    d.halo_ns = A.get<int>((size_t)d.n_regblk);
    d.tile_list = A.get<int>((size_t)d.n_regblk);
    d.s_rec = A.get<SpringRec>(d.use_lds ? nnz_s : 1);
    d.d_rec = A.get<DamperRec>(d.use_lds ? nnz_d : 1);
    const size_t us = d.use_lds ? 1 : nnz_s, ud = d.use_lds ? 1 : nnz_d;     // unpacked arrays: fallback path only
    d.s_other = A.get<int>(us); d.s_d0 = A.get<float>(us); d.s_meta = A.get<int>(us);
    d.d_o0 = A.get<int>(ud); d.d_o1 = A.get<int>(ud); d.d_o2 = A.get<int>(ud);
    d.d_w = A.get<float>(nnz_d); d.d_meta = A.get<int>(ud);
    for (int s = 0; s < 2; ++s) { d.pose[s] = A.get<Pose>(K); d.xl[s] = A.get<double>(3 * nr); }
    d.pose_init = A.get<Pose>(K);
    d.xl_init = A.get<double>(3 * nr);
    d.D = A.get<double>(6 * nr);
    d.Hpl = A.get<double>(d.use_lds ? 1 : 18 * nr);
    d.rowrec = A.get<RowRec>(d.use_lds ? nr : 1);
    d.s_g = A.get<double>(3 * us);
    d.d_s = A.get<double>(ud);
    d.Hpp = A.get<double>(21 * K);
    d.bp = A.get<double>(6 * K);
    d.bl = A.get<double>(3 * nr);
    d.Dinv = A.get<double>(6 * nr);
    d.Hppinv = A.get<double>(36 * K);
    double** pv[] = {&d.xp, &d.rp, &d.up, &d.pp, &d.sp, &d.wp};
    for (auto p : pv) *p = A.get<double>(6 * K);
    double** rvv[] = {&d.xv, &d.rv, &d.uv3, &d.pv, &d.sv, &d.wv};
    for (auto p : rvv) *p = A.get<double>(3 * nr);
    d.rp2 = A.get<double>(6 * K); d.sp2 = A.get<double>(6 * K); d.up2 = A.get<double>(6 * K);
    d.rv2 = A.get<double>(d.fused ? 3 * nr : 1); d.sv2 = A.get<double>(d.fused ? 3 * nr : 1); d.wv2 = A.get<double>(d.fused ? 3 * nr : 1);
    d.part_spmv2 = A.get<double>(d.fused ? NPART * (size_t)d.n_regblk : 1);
    {
        const size_t nb = d.coarse ? (size_t)d.n_regblk : 1, nc = d.coarse ? (size_t)d.co_n : 1;
        d.co_ct = A.get<double>(nb * (d.coarse ? (size_t)d.n_groups : 1) * 6);
        d.co_cp = A.get<double>(nb * 18);
        d.co_tb = A.get<double>(nb * 4);
        d.co_bt = A.get<double>(nb * 6);
        d.co_bti = A.get<double>(nb * 6);
        d.part_ts = A.get<double>(nb * 9); d.part_ts2 = A.get<double>(nb * 9);
        d.co_c0 = A.get<double>(nc * nc); d.co_nn = A.get<double>(nc); d.co_bc = A.get<double>(nc);
        d.co_inv = A.get<double>(nc * nc); d.co_y0 = A.get<double>(nc);
    }
    d.tile_desc = A.get<int>(d.fused ? 8 * (size_t)d.n_regblk : 4);
    d.halo_fix = A.get<int>(d.fused ? BLK * (size_t)d.n_regblk : 4);
    d.red = A.get<double>(3 + 6 * K);
    d.part_lin = A.get<double>(32 * (size_t)d.n_groups);
    d.part_reg = A.get<double>(2 * (size_t)d.n_regblk);
    d.part_spmv = A.get<double>(NPART * (size_t)d.n_regblk);
    d.part_apply = A.get<double>((size_t)d.n_vecblk);
    d.scal = A.get<double>(SC_N);
    d.flags = A.get<int>(8);
    e->t_vrow = A.get<int>(d.M);
    e->t_sp = A.get<int>(2 * (size_t)d.n_sp);
    e->t_dm = A.get<int>(4 * (size_t)d.n_dm);
    e->t_d0 = A.get<float>(d.n_sp);
    e->t_w = A.get<float>(d.n_dm);
    e->t_out = A.get<double>(2 * (size_t)d.M + (size_t)d.n_sp + 3 * (size_t)d.n_dm);
}

template <class Tp>
static int h2d(nrs_ctx* c, Tp* dst, const std::vector<Tp>& src) {
    if (!src.empty()) NRS_HIP(c, hipMemcpyAsync(dst, src.data(), sizeof(Tp) * src.size(), hipMemcpyHostToDevice, c->stream));
    return NRS_OK;
}

// per-edge masks -> per-incidence meta words and per-row flags (host), then upload
static int push_masks(nrs_ctx* c, Engine* e, const uint8_t* sp_active, const uint8_t* dm_active) {
    Dev& d = e->d;
    auto vfixed = [&](int v) { return (e->h_rflag[e->vrow[v]] & RF_FIXED) != 0; };
    for (int s = 0; s < d.n_sp; ++s) {
        const int i = e->sp_ij[2 * s], j = e->sp_ij[2 * s + 1];
        const bool act = (!sp_active || sp_active[s]) && !(vfixed(i) && vfixed(j));
        const int m = act ? SM_ACTIVE : 0;
        e->h_s_meta[e->sp_pos[2 * s]] = m | (act ? SM_COUNT : 0);
        e->h_s_meta[e->sp_pos[2 * s + 1]] = m;
    }
    for (int s = 0; s < d.n_dm; ++s) {
        bool allfix = true;
        int first = -1;
        for (int r = 0; r < 4; ++r) {
            const int v = e->dm_idx[4 * s + r];
            if (v >= 0) { if (first < 0) first = r; allfix = allfix && vfixed(v); }
        }
        const bool act = (!dm_active || dm_active[s]) && !allfix;
        for (int r = 0; r < 4; ++r) {
            const int p = e->dm_pos[4 * s + r];
            if (p < 0) continue;
            e->h_d_meta[p] = r | (act ? DM_ACTIVE : 0) | ((act && r == first) ? DM_COUNT : 0);
        }
    }
    for (int s = 0; s < d.n_un; ++s) {
        const bool act = !vfixed(e->un_ij[2 * s]);
        e->h_d_meta[e->un_pos[s]] = 2 | DM_UNARY | (act ? (DM_ACTIVE | DM_COUNT) : 0);
    }
    if (d.use_lds) {
        for (size_t i = 0; i < e->h_s_rec.size(); ++i) {
            const int m = e->h_s_meta[i];
            e->h_s_rec[i].meta = (uint16_t)(((m & SM_ACTIVE) ? SR_ACTIVE : 0) | ((m & SM_COUNT) ? SR_COUNT : 0));
        }
        for (size_t i = 0; i < e->h_d_rec.size(); ++i)
            e->h_d_rec[i].meta = e->h_d_meta[i] < 0 ? REC_NONE : (uint16_t)e->h_d_meta[i];
        NRS_TRY(h2d(c, d.s_rec, e->h_s_rec));
        NRS_TRY(h2d(c, d.d_rec, e->h_d_rec));
    } else {
        NRS_TRY(h2d(c, d.s_meta, e->h_s_meta));
        NRS_TRY(h2d(c, d.d_meta, e->h_d_meta));
    }
    NRS_TRY(h2d(c, d.rflag, e->h_rflag));
    NRS_TRY(h2d(c, d.pose_fixed, e->h_pose_fixed));
    return NRS_OK;
}

int engine_create(nrs_ctx* c, const EngineSpec& s, Arena* arena, Engine** out) {
    *out = nullptr;
    if (s.K <= 0 || s.M <= 0 || !s.poses || !s.x || !s.lm_pose || !s.uv || !s.rflag || s.n_sp < 0 || s.n_dm < 0 || s.n_un < 0)
        return c->fail(NRS_ERR_INVALID, "engine: bad specification");
    for (int i = 0; i < s.M; ++i)
        if (s.lm_pose[i] < 0 || s.lm_pose[i] >= s.K || (i > 0 && s.lm_pose[i] < s.lm_pose[i - 1]))
            return c->fail(NRS_ERR_INVALID, "vertex pose index must be non-decreasing and in [0, n_poses)");
    for (int64_t i = 0; i < 2 * (int64_t)s.n_sp; ++i)
        if (s.sp_ij[i] < 0 || s.sp_ij[i] >= s.M) return c->fail(NRS_ERR_INVALID, "spring index out of range");
    for (int64_t i = 0; i < 4 * (int64_t)s.n_dm; ++i)
        if (s.dm_idx[i] < -1 || s.dm_idx[i] >= s.M) return c->fail(NRS_ERR_INVALID, "damper index out of range");
    for (int64_t i = 0; i < 2 * (int64_t)s.n_un; ++i)
        if (s.un_ij[i] < 0 || s.un_ij[i] >= s.M) return c->fail(NRS_ERR_INVALID, "unary damper index out of range");
    const bool tm = getenv("NRS_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!tm) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[nrs] engine_create %-18s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    NRS_HIP(c, hipSetDevice(c->device));
    Engine* e = new (std::nothrow) Engine();
    if (!e) return c->fail(NRS_ERR_ALLOC, "out of host memory");
    struct Guard { nrs_ctx* c; Engine* e; bool keep = false; ~Guard() { if (!keep) engine_destroy(c, e); } } guard{c, e};
    e->arena = arena;
    Dev& d = e->d;
    memset(&d, 0, sizeof(d));
    // lanes per row: 2 measured best on C2 (92k rows), 8 on single-frame problems (4.5k rows), where
    // the kernels are bound by per-lane latency chains rather than by traffic (profiles/README.md)
    int n_pad_rows = 0;
    {
        std::vector<int> cnt(s.K, 0);
        for (int i = 0; i < s.M; ++i) cnt[s.lm_pose[i]]++;
        for (int k = 0; k < s.K; ++k) n_pad_rows += std::max(1, (cnt[k] + ROW_ALIGN - 1) / ROW_ALIGN) * ROW_ALIGN;
    }
    int T = n_pad_rows >= 32768 ? 2 : 8;
    if (const char* ev = getenv("NRS_SELL_T")) {
        const int v = atoi(ev);
        if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) T = v;
    }
    d.T = T;
    d.K = s.K; d.M = s.M; d.n_sp = s.n_sp; d.n_dm = s.n_dm; d.n_un = s.n_un;
    d.cam = s.cam;
    d.info_reproj = s.info_reproj; d.delta_reproj = s.delta_reproj;
    d.info_pos = s.info_pos; d.delta_pos = s.delta_pos;
    d.info_spatial = s.info_spatial; d.delta_spatial = s.delta_spatial;
    d.k_spring = s.k_spring; d.spring_form = s.spring_form;

    // ---- row layout: pose-major, each pose padded to ROW_ALIGN rows, Morton order inside
    std::vector<int> pose_ptr(s.K + 1, 0);
    for (int i = 0; i < s.M; ++i) pose_ptr[s.lm_pose[i] + 1]++;
    for (int k = 0; k < s.K; ++k) pose_ptr[k + 1] += pose_ptr[k];
    std::vector<int> pose_grp_ptr(s.K + 1, 0), grp_pose;
    for (int k = 0; k < s.K; ++k) {
        const int n = pose_ptr[k + 1] - pose_ptr[k];
        const int ng = std::max(1, (n + ROW_ALIGN - 1) / ROW_ALIGN);
        pose_grp_ptr[k + 1] = pose_grp_ptr[k] + ng;
        for (int g = 0; g < ng; ++g) grp_pose.push_back(k);
    }
    d.n_groups = pose_grp_ptr[s.K];
    d.n_rows = d.n_groups * ROW_ALIGN;
    d.n_regblk = d.n_rows / (BLK / T);
    d.n_vecblk = d.n_rows / BLK;
    e->vrow.resize(s.M);
    {
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        auto pos = [&](int v, int a) { return s.x[3 * (size_t)v + a] + (s.X0 ? s.X0[3 * (size_t)v + a] : 0.0); };
        for (int v = 0; v < s.M; ++v)
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], pos(v, a)); hi[a] = std::max(hi[a], pos(v, a)); }
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
        for (int k = 0; k < s.K; ++k) {
            keys.clear();
            for (int v = pose_ptr[k]; v < pose_ptr[k + 1]; ++v) {
                uint64_t code = 0;
                if (morton)
                    for (int a = 0; a < 3; ++a) {
                        const double ext = hi[a] - lo[a];
                        const double f = ext > 0 ? (pos(v, a) - lo[a]) / ext : 0.0;
                        code |= spread((uint64_t)(f * 2097151.0)) << a;
                    }
                keys.emplace_back(code, v);
            }
            std::stable_sort(keys.begin(), keys.end());
            for (size_t i = 0; i < keys.size(); ++i) e->vrow[keys[i].second] = pose_grp_ptr[k] * ROW_ALIGN + (int)i;
        }
    }
    mark("row layout");
    // ---- incidence lists -> sliced ELL, built with two counting passes (no per-row containers)
    const int Rw = 64 / T;
    const int n_slices = d.n_rows / Rw;
    const int dm_slots = 4 * s.n_dm;
    e->sp_pos.assign(2 * (size_t)s.n_sp, -1);
    e->dm_pos.assign(4 * (size_t)s.n_dm, -1);
    e->un_pos.assign((size_t)s.n_un, -1);
    std::vector<int> cnt_s(d.n_rows, 0), cnt_d(d.n_rows, 0);
    for (int q = 0; q < s.n_sp; ++q) { cnt_s[e->vrow[s.sp_ij[2 * q]]]++; cnt_s[e->vrow[s.sp_ij[2 * q + 1]]]++; }
    for (int64_t q = 0; q < 4 * (int64_t)s.n_dm; ++q)
        if (s.dm_idx[q] >= 0) cnt_d[e->vrow[s.dm_idx[q]]]++;
    for (int q = 0; q < s.n_un; ++q) cnt_d[e->vrow[s.un_ij[2 * q]]]++;
    std::vector<int> ss_ptr(n_slices + 1, 0), sd_ptr(n_slices + 1, 0);
    for (int sl = 0; sl < n_slices; ++sl) {
        int ws = 0, wd = 0;
        for (int r = 0; r < Rw; ++r) {
            ws = std::max(ws, (cnt_s[sl * Rw + r] + T - 1) / T);
            wd = std::max(wd, (cnt_d[sl * Rw + r] + T - 1) / T);
        }
        ss_ptr[sl + 1] = ss_ptr[sl] + ws * 64;
        sd_ptr[sl + 1] = sd_ptr[sl] + wd * 64;
    }
    const size_t nnz_s = (size_t)ss_ptr[n_slices], nnz_d = (size_t)sd_ptr[n_slices];
    d.ss_nnz = (int)nnz_s;
    d.sd_nnz = (int)nnz_d;
    // packed position of the k-th incidence of a row
    auto pos_of = [&](const std::vector<int>& ptr, int row, int k) {
        const int sl = row / Rw, r = row - sl * Rw;
        return (size_t)ptr[sl] + (size_t)(k / T) * 64 + (size_t)r * T + (size_t)(k % T);
    };
    std::vector<int> S_other(nnz_s, -1), D_o(3 * nnz_d, -1), D_role(nnz_d, -1);
    std::vector<float> S_d0(nnz_s, 0.f), D_w(nnz_d, 0.f);
    std::fill(cnt_s.begin(), cnt_s.end(), 0);
    std::fill(cnt_d.begin(), cnt_d.end(), 0);
    for (int q = 0; q < s.n_sp; ++q) {
        const int a = e->vrow[s.sp_ij[2 * q]], b = e->vrow[s.sp_ij[2 * q + 1]];
        const size_t pa = pos_of(ss_ptr, a, cnt_s[a]++), pb = pos_of(ss_ptr, b, cnt_s[b]++);
        S_other[pa] = b; S_d0[pa] = s.sp_d0[q]; e->sp_pos[2 * (size_t)q] = (int)pa;
        S_other[pb] = a; S_d0[pb] = s.sp_d0[q]; e->sp_pos[2 * (size_t)q + 1] = (int)pb;
    }
    for (int q = 0; q < s.n_dm; ++q) {
        int r4[4];
        for (int k = 0; k < 4; ++k) r4[k] = s.dm_idx[4 * q + k] >= 0 ? e->vrow[s.dm_idx[4 * q + k]] : -1;
        for (int role = 0; role < 4; ++role) {
            if (r4[role] < 0) continue;
            const size_t pz = pos_of(sd_ptr, r4[role], cnt_d[r4[role]]++);
            int z = 0;
            for (int k = 0; k < 4; ++k)
                if (k != role) D_o[3 * pz + z++] = r4[k];
            D_w[pz] = s.dm_w[q];
            D_role[pz] = role;
            e->dm_pos[4 * (size_t)q + role] = (int)pz;
        }
    }
    for (int q = 0; q < s.n_un; ++q) {           // own role 2 (1n, +), value-only other in role 3 (2n, -)
        const int row = e->vrow[s.un_ij[2 * q]];
        const size_t pz = pos_of(sd_ptr, row, cnt_d[row]++);
        D_o[3 * pz + 2] = e->vrow[s.un_ij[2 * q + 1]];
        D_w[pz] = s.un_w[q];
        D_role[pz] = 2;
        e->un_pos[q] = (int)pz;
    }
    (void)dm_slots;
    mark("sell pack");
    // ---- LDS staging: per workgroup (= 4 slices = BLK/T rows) the sorted list of rows referenced
    // outside the tile; neighbour ids become tile-local
    d.tile_rows = BLK / T;
    std::vector<int> halo_ptr(d.n_regblk + 1, 0), halo_rows, halo_ns(d.n_regblk, 0);
    d.max_halo_s = 0;
    std::vector<int> L_s(nnz_s, -1), L_d(3 * nnz_d, -1);      // tile-local ids
    {
        // tiles are independent: a few host threads each take a contiguous range of tiles
        const int nt = std::max(1, std::min({8, (int)std::thread::hardware_concurrency(), d.n_regblk / 128}));
        std::vector<std::vector<int>> part(nt);
        std::vector<int> cnt(d.n_regblk, 0);
        auto work = [&](int ti) {
            const int b0 = (int)((int64_t)d.n_regblk * ti / nt), b1 = (int)((int64_t)d.n_regblk * (ti + 1) / nt);
            std::vector<int> stamp(d.n_rows, -1), local(d.n_rows, 0), ext;
            for (int b = b0; b < b1; ++b) {
                const int row0 = b * d.tile_rows, row1 = row0 + d.tile_rows;
                ext.clear();
                const size_t s0 = (size_t)ss_ptr[b * 4], s1 = (size_t)ss_ptr[b * 4 + 4];
                const size_t d0 = (size_t)sd_ptr[b * 4], d1 = (size_t)sd_ptr[b * 4 + 4];
                auto see = [&](int o) {
                    if (o >= 0 && (o < row0 || o >= row1) && stamp[o] != b) { stamp[o] = b; ext.push_back(o); }
                };
                // spring neighbours first (the SpMV stages positions for them only), then damper-only rows
                for (size_t p2 = s0; p2 < s1; ++p2) see(S_other[p2]);
                const size_t ns = ext.size();
                for (size_t p2 = 3 * d0; p2 < 3 * d1; ++p2) see(D_o[p2]);
                std::sort(ext.begin(), ext.begin() + ns);
                std::sort(ext.begin() + ns, ext.end());
                halo_ns[b] = (int)ns;
                for (size_t i = 0; i < ext.size(); ++i) local[ext[i]] = d.tile_rows + (int)i;
                auto loc = [&](int o) { return o < 0 ? -1 : (o >= row0 && o < row1) ? o - row0 : local[o]; };
                for (size_t p2 = s0; p2 < s1; ++p2) L_s[p2] = loc(S_other[p2]);
                for (size_t p2 = 3 * d0; p2 < 3 * d1; ++p2) L_d[p2] = loc(D_o[p2]);
                part[ti].insert(part[ti].end(), ext.begin(), ext.end());
                cnt[b] = (int)ext.size();
            }
        };
        if (nt == 1) work(0);
        else {
            std::vector<std::thread> th;
            for (int ti = 0; ti < nt; ++ti) th.emplace_back(work, ti);
            for (auto& t : th) t.join();
        }
        for (int b = 0; b < d.n_regblk; ++b) {
            halo_ptr[b + 1] = halo_ptr[b] + cnt[b];
            d.max_halo = std::max(d.max_halo, cnt[b]);
            d.max_halo_s = std::max(d.max_halo_s, halo_ns[b]);
        }
        halo_rows.reserve((size_t)halo_ptr[d.n_regblk]);
        for (int ti = 0; ti < nt; ++ti) halo_rows.insert(halo_rows.end(), part[ti].begin(), part[ti].end());
    }
    // ---- tile classes: if a few tiles have much larger halos than the rest they get their own
    // launch (class 1) with their own LDS size, and the bulk (class 0) keeps its occupancy
    std::vector<int> tile_list(d.n_regblk);
    {
        std::vector<int> hs(d.n_regblk);
        for (int b = 0; b < d.n_regblk; ++b) hs[b] = halo_ptr[b + 1] - halo_ptr[b];
        std::vector<int> sorted = hs;
        std::sort(sorted.begin(), sorted.end());
        int cut = d.max_halo;
        if (d.n_regblk >= 1024) {                                  // small problems are latency-bound: one launch
            // (the second launch has to fill the chip by itself: >= 4 workgroups per CU, or be needed
            // for the bulk to fit the LDS budget at all)
            const int p97 = sorted[(size_t)(0.97 * (d.n_regblk - 1))];
            const bool fits = sizeof(double) * 3 * (size_t)(2 * d.tile_rows + d.max_halo + d.max_halo_s + 2) <= 48 * 1024;
            if (4 * d.max_halo > 5 * p97 && (d.n_regblk - (int)(0.97 * d.n_regblk) >= 1024 || !fits) && !getenv("NRS_ONE_CLASS")) cut = p97;
        }
        if (getenv("NRS_TILE_CUT_PCT")) {                          // test switch: force a split at a percentile
            const double pct = atof(getenv("NRS_TILE_CUT_PCT")) / 100.0;
            cut = sorted[(size_t)(pct * (d.n_regblk - 1))];
        }
        int n0 = 0;
        for (int b = 0; b < d.n_regblk; ++b) if (hs[b] <= cut) tile_list[n0++] = b;
        int n1 = n0;
        for (int b = 0; b < d.n_regblk; ++b) if (hs[b] > cut) tile_list[n1++] = b;
        d.n_tiles_cls[0] = n0; d.n_tiles_cls[1] = d.n_regblk - n0;
        d.cap_h[0] = d.cap_h[1] = d.cap_s[0] = d.cap_s[1] = 0;
        for (int b = 0; b < d.n_regblk; ++b) {
            const int cls = hs[b] <= cut ? 0 : 1;
            d.cap_h[cls] = std::max(d.cap_h[cls], hs[b]);
            d.cap_s[cls] = std::max(d.cap_s[cls], halo_ns[b]);
        }
    }
    d.use_lds = 1;
    size_t lds_need = 0;
    for (int cls = 0; cls < 2; ++cls) {
        if (!d.n_tiles_cls[cls]) continue;
        lds_need = std::max(lds_need, sizeof(double) * 3 * (size_t)(d.tile_rows + d.cap_h[cls]) * (s.X0 ? 2 : 1));                          // linearise
        lds_need = std::max(lds_need, sizeof(double) * 3 * (size_t)(2 * d.tile_rows + d.cap_h[cls] + d.cap_s[cls] + 2));                    // operator: u + positions
    }
    if (getenv("NRS_NO_LDS") || lds_need > 64 * 1024 - 512 || d.tile_rows + d.max_halo >= 65535) d.use_lds = 0;   // irregular graph / A-B switch
    // single-launch PCG iteration for problems that are bound by launch latency, not by traffic
    const int fused_max = getenv("NRS_FUSED_MAX_ROWS") ? atoi(getenv("NRS_FUSED_MAX_ROWS")) : 32768;
    d.fused = (d.use_lds && d.n_rows < fused_max && !getenv("NRS_NO_FUSED")) ? 1 : 0;
    d.hier = (d.n_regblk > 4096 || getenv("NRS_HIER")) ? 1 : 0;
    // two-level preconditioner: fused path, one pose, small enough coarse system
    d.co_n = 3 * d.n_groups + 6;
    // (worth its per-iteration cost on the pose + deformation problems; the lost-point stage, pose
    // fixed and few free rows, converges in a few dozen block-Jacobi iterations anyway)
    const bool pose_free = !(s.pose_fixed && s.pose_fixed[0]);
    const size_t fused_shm = sizeof(double) * (6 * (size_t)(d.tile_rows + d.max_halo) + 9 * (size_t)d.n_regblk + 10 * CO_MAX);
    d.coarse = (d.fused && s.K == 1 && pose_free && d.co_n <= CO_MAX && d.n_regblk <= BLK && fused_shm <= 63 * 1024 &&
                !getenv("NRS_NO_COARSE")) ? 1 : 0;
    mark("halo");
    if (tm) fprintf(stderr, "[nrs] tiles %d x %d rows (T=%d), halo rows: max %d, mean %.1f, spring part max %d, classes %d (cap %d/%d) + %d (cap %d/%d), lds %d, fused %d\n", d.n_regblk, d.tile_rows, T, d.max_halo, (double)halo_rows.size() / d.n_regblk, d.max_halo_s, d.n_tiles_cls[0], d.cap_h[0], d.cap_s[0], d.n_tiles_cls[1], d.cap_h[1], d.cap_s[1], d.use_lds, d.fused);
    if (tm) fprintf(stderr, "[nrs] coarse level: wanted %d (fused %d, K %d, unknowns %d <= %d), enabled %d\n", d.fused && s.K == 1, d.fused, s.K, 3 * d.n_groups + 6, CO_MAX, d.coarse);
    // ---- device memory: one arena allocation, reused across calls when large enough
    ArenaPlan dry{arena, true};
    {
        Dev tmp = d;
        Engine te;
        carve(dry, tmp, s.X0 != nullptr, nnz_s, nnz_d, ss_ptr.size() - 1, halo_rows.size(), &te);
    }
    if (dry.off > arena->cap) {
        NRS_HIP(c, hipStreamSynchronize(c->stream));
        arena_release(arena);
        const size_t want = dry.off + dry.off / 8;
        hipError_t he = hipMalloc((void**)&arena->base, want);
        if (he != hipSuccess) return c->fail(NRS_ERR_ALLOC, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(he));
        arena->cap = want;
    }
    ArenaPlan real{arena, false};
    carve(real, d, s.X0 != nullptr, nnz_s, nnz_d, ss_ptr.size() - 1, halo_rows.size(), e);

    mark("arena");
    // ---- host mirrors + uploads
    e->sp_ij.assign(s.sp_ij, s.sp_ij + 2 * (size_t)s.n_sp);
    e->sp_d0.assign(s.sp_d0, s.sp_d0 + (size_t)s.n_sp);
    e->dm_idx.assign(s.dm_idx, s.dm_idx + 4 * (size_t)s.n_dm);
    e->dm_w.assign(s.dm_w, s.dm_w + (size_t)s.n_dm);
    e->un_ij.assign(s.un_ij, s.un_ij + 2 * (size_t)s.n_un);
    e->un_w.assign(s.un_w, s.un_w + (size_t)s.n_un);
    e->h_rflag.assign(d.n_rows, RF_FIXED);            // padding rows: no edges, never move
    for (int v = 0; v < s.M; ++v) e->h_rflag[e->vrow[v]] = s.rflag[v];
    e->h_pose_fixed.assign(s.K, 0);
    if (s.pose_fixed) e->h_pose_fixed.assign(s.pose_fixed, s.pose_fixed + s.K);
    std::vector<float> uv((size_t)d.n_rows * 2, 0.f);
    std::vector<double> xl((size_t)d.n_rows * 3, 0.0), X0;
    if (s.X0) X0.assign((size_t)d.n_rows * 3, 0.0);
    for (int v = 0; v < s.M; ++v) {
        const size_t row = (size_t)e->vrow[v];
        uv[2 * row] = s.uv[2 * v];
        uv[2 * row + 1] = s.uv[2 * v + 1];
        for (int k = 0; k < 3; ++k) {
            xl[3 * row + k] = s.x[3 * (size_t)v + k];
            if (s.X0) X0[3 * row + k] = s.X0[3 * (size_t)v + k];
        }
    }
    e->h_s_meta.assign(nnz_s, 0);
    e->h_d_meta.assign(nnz_d, -1);
    for (size_t i = 0; i < nnz_d; ++i)
        if (D_role[i] >= 0) e->h_d_meta[i] = D_role[i];
    std::vector<int> d_o0, d_o1, d_o2;
    if (d.use_lds) {
        auto u16 = [](int v) { return v < 0 ? REC_NONE : (uint16_t)v; };
        e->h_s_rec.resize(nnz_s);
        for (size_t i = 0; i < nnz_s; ++i) {
            SpringRec& r = e->h_s_rec[i];
            r.qc = 0;
            r.other = u16(L_s[i]); r.meta = 0; r.d0 = S_d0[i];
        }
        e->h_d_rec.resize(nnz_d);
        for (size_t i = 0; i < nnz_d; ++i) {
            DamperRec& r = e->h_d_rec[i];
            r.o0 = u16(L_d[3 * i]); r.o1 = u16(L_d[3 * i + 1]); r.o2 = u16(L_d[3 * i + 2]);
            r.meta = REC_NONE; r.s = 0;
        }
    } else {
        d_o0.resize(nnz_d); d_o1.resize(nnz_d); d_o2.resize(nnz_d);
        for (size_t i = 0; i < nnz_d; ++i) { d_o0[i] = D_o[3 * i]; d_o1[i] = D_o[3 * i + 1]; d_o2[i] = D_o[3 * i + 2]; }
    }
    const std::vector<int>& s_other = S_other;
    const std::vector<float>& s_d0 = S_d0;
    const std::vector<float>& d_w = D_w;
    mark("host mirrors");
    std::vector<Pose> poses(s.poses, s.poses + s.K);
    NRS_TRY(h2d(c, d.grp_pose, grp_pose));
    NRS_TRY(h2d(c, d.pose_grp_ptr, pose_grp_ptr));
    NRS_TRY(h2d(c, d.uv, uv));
    NRS_TRY(h2d(c, d.xl_init, xl));
    if (s.X0) NRS_TRY(h2d(c, d.X0, X0));
    NRS_TRY(h2d(c, d.pose_init, poses));
    NRS_TRY(h2d(c, d.ss_ptr, ss_ptr));
    NRS_TRY(h2d(c, d.sd_ptr, sd_ptr));
    NRS_TRY(h2d(c, d.halo_ptr, halo_ptr));
    NRS_TRY(h2d(c, d.halo_rows, halo_rows));
    NRS_TRY(h2d(c, d.halo_ns, halo_ns));
    NRS_TRY(h2d(c, d.tile_list, tile_list));
    if (d.fused) {
        std::vector<int> tile_desc(8 * (size_t)d.n_regblk, 0), halo_fix((size_t)BLK * d.n_regblk, 0);
        const int rb = ROW_ALIGN / d.tile_rows;
        for (int b = 0; b < d.n_regblk; ++b) {
            const int kf = grp_pose[(size_t)b * d.tile_rows / ROW_ALIGN];
            int* td = &tile_desc[8 * (size_t)b];
            td[0] = kf; td[1] = pose_grp_ptr[kf] * rb; td[2] = pose_grp_ptr[kf + 1] * rb;
            td[3] = halo_ptr[b]; td[4] = halo_ptr[b + 1] - halo_ptr[b];
            for (int i = 0; i < td[4] && i < BLK; ++i) halo_fix[(size_t)b * BLK + i] = halo_rows[td[3] + i];
        }
        NRS_TRY(h2d(c, d.tile_desc, tile_desc));
        NRS_TRY(h2d(c, d.halo_fix, halo_fix));
    }
    if (!d.use_lds) {
        NRS_TRY(h2d(c, d.s_other, s_other));
        NRS_TRY(h2d(c, d.s_d0, s_d0));
        NRS_TRY(h2d(c, d.d_o0, d_o0));
        NRS_TRY(h2d(c, d.d_o1, d_o1));
        NRS_TRY(h2d(c, d.d_o2, d_o2));
    }
    NRS_TRY(h2d(c, d.d_w, d_w));
    NRS_TRY(push_masks(c, e, s.sp_active, s.dm_active));
    NRS_TRY(h2d(c, e->t_vrow, e->vrow));
    NRS_TRY(h2d(c, e->t_sp, e->sp_ij));
    NRS_TRY(h2d(c, e->t_dm, e->dm_idx));
    NRS_TRY(h2d(c, e->t_d0, e->sp_d0));
    NRS_TRY(h2d(c, e->t_w, e->dm_w));
    NRS_HIP(c, hipMemsetAsync(d.part_apply, 0, sizeof(double) * (size_t)d.n_vecblk, c->stream));
    NRS_HIP(c, hipMemsetAsync(d.scal, 0, sizeof(double) * SC_N, c->stream));
    NRS_HIP(c, hipMemsetAsync(d.flags, 0, sizeof(int) * 8, c->stream));
    mark("uploads enqueued");
    if (!c->pin_scal) NRS_HIP(c, hipHostMalloc((void**)&c->pin_scal, sizeof(double) * SC_N, hipHostMallocMapped));      // pinned mirrors live in
    if (!c->pin_flags) NRS_HIP(c, hipHostMalloc((void**)&c->pin_flags, sizeof(int) * 8, hipHostMallocMapped));         // the context (reused)
    e->h_scal = c->pin_scal;
    e->h_flags = c->pin_flags;
    e->d.h_scal = c->pin_scal;          // hipHostMalloc memory is mapped: same pointer on the device
    e->d.h_flags = c->pin_flags;
    NRS_HIP(c, hipStreamSynchronize(c->stream));       // host staging vectors die here
    mark("pinned+sync");
    NRS_TRY(engine_reset(c, e));
    guard.keep = true;
    *out = e;
    return NRS_OK;
}

void engine_destroy(nrs_ctx* c, Engine* e) {
    if (!e) return;
    (void)hipStreamSynchronize(c->stream);
    delete e;
}

int engine_update_flags(nrs_ctx* c, Engine* e, const uint8_t* rflag, const uint8_t* pose_fixed,
                        const uint8_t* sp_active, const uint8_t* dm_active) {
    if (rflag)
        for (int v = 0; v < e->d.M; ++v) e->h_rflag[e->vrow[v]] = rflag[v];
    if (pose_fixed) e->h_pose_fixed.assign(pose_fixed, pose_fixed + e->d.K);
    NRS_TRY(push_masks(c, e, sp_active, dm_active));
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    return NRS_OK;
}

int engine_reset(nrs_ctx* c, Engine* e) {
    Dev& d = e->d;
    e->cur = 0;
    e->pred_iters = 0; e->pred_peek = 0; e->first_trial_accepted = false;   // batch-size predictors start fresh, as in a new engine
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

template <bool LIN, bool LDS>
static void launch_reg2(nrs_ctx* c, const Dev& d, const double* xl, size_t shm, int n, int cls) {
    const dim3 g(((n + 7) / 8) * 8), b(BLK);
    switch (d.T) {
        case 1: hipLaunchKernelGGL((k_reg<1, LIN, LDS>), g, b, shm, c->stream, d, xl, cls); break;
        case 4: hipLaunchKernelGGL((k_reg<4, LIN, LDS>), g, b, shm, c->stream, d, xl, cls); break;
        case 8: hipLaunchKernelGGL((k_reg<8, LIN, LDS>), g, b, shm, c->stream, d, xl, cls); break;
        case 16: hipLaunchKernelGGL((k_reg<16, LIN, LDS>), g, b, shm, c->stream, d, xl, cls); break;
        default: hipLaunchKernelGGL((k_reg<2, LIN, LDS>), g, b, shm, c->stream, d, xl, cls); break;
    }
}

template <bool LIN>
static void launch_reg(nrs_ctx* c, const Dev& d, const double* xl) {
    if (!d.use_lds) { launch_reg2<LIN, false>(c, d, xl, 0, d.n_regblk, 0); return; }
    for (int cls = 0; cls < 2; ++cls) {
        if (d.n_tiles_cls[cls] == 0) continue;
        const size_t shm = sizeof(double) * 3 * (size_t)(d.tile_rows + d.cap_h[cls]) * (d.X0 ? 2 : 1);
        launch_reg2<LIN, true>(c, d, xl, shm, d.n_tiles_cls[cls], cls);
    }
}

template <bool LDS>
static void launch_spmv2(nrs_ctx* c, const Dev& d, double lam, size_t shm, int it) {
    const dim3 g(((d.n_regblk + 7) / 8) * 8), b(BLK);
    switch (d.T) {
        case 1: hipLaunchKernelGGL((k_spmv<1, LDS>), g, b, shm, c->stream, d, lam, it); break;
        case 4: hipLaunchKernelGGL((k_spmv<4, LDS>), g, b, shm, c->stream, d, lam, it); break;
        case 8: hipLaunchKernelGGL((k_spmv<8, LDS>), g, b, shm, c->stream, d, lam, it); break;
        case 16: hipLaunchKernelGGL((k_spmv<16, LDS>), g, b, shm, c->stream, d, lam, it); break;
        default: hipLaunchKernelGGL((k_spmv<2, LDS>), g, b, shm, c->stream, d, lam, it); break;
    }
}

static void launch_spmv(nrs_ctx* c, const Dev& d, double lam, int it) {
    if (!d.use_lds) { launch_spmv2<false>(c, d, lam, 0, it); return; }
    for (int cls = 0; cls < 2; ++cls) {
        const int n = d.n_tiles_cls[cls];
        if (n == 0) continue;
        const size_t shm = sizeof(double) * 3 * (size_t)(2 * d.tile_rows + d.cap_h[cls] + d.cap_s[cls] + 2);
        const dim3 g(((n + 7) / 8) * 8), b(BLK);
        switch (d.T) {
            case 1: hipLaunchKernelGGL((k_spmv_f<1>), g, b, shm, c->stream, d, lam, cls, it); break;
            case 4: hipLaunchKernelGGL((k_spmv_f<4>), g, b, shm, c->stream, d, lam, cls, it); break;
            case 8: hipLaunchKernelGGL((k_spmv_f<8>), g, b, shm, c->stream, d, lam, cls, it); break;
            case 16: hipLaunchKernelGGL((k_spmv_f<16>), g, b, shm, c->stream, d, lam, cls, it); break;
            default: hipLaunchKernelGGL((k_spmv_f<2>), g, b, shm, c->stream, d, lam, cls, it); break;
        }
    }
}

// errors (+ linearisation) at a given state; leaves chi2 (and max diag) in scal[]
template <bool LIN>
static int evaluate(nrs_ctx* c, Engine* e, int which) {
    if (LIN) { e->d.lin_pose = e->d.pose[which]; e->d.lin_xl = e->d.xl[which]; }   // the PCG kernels re-form factors from it
    const Dev& d = e->d;
    const dim3 gg(((d.n_groups + 7) / 8) * 8), b(BLK);
    if (LIN) {
        Timer t(c, &c->prof.linearize_ms, &c->prof.linearize_launches);
        hipLaunchKernelGGL((k_reproj<LIN>), gg, b, 0, c->stream, d, d.pose[which], d.xl[which]);
        launch_reg<LIN>(c, d, d.xl[which]);
    } else {
        hipLaunchKernelGGL((k_reproj<LIN>), gg, b, 0, c->stream, d, d.pose[which], d.xl[which]);
        launch_reg<LIN>(c, d, d.xl[which]);
    }
    if (LIN) hipLaunchKernelGGL(k_pose_sums, dim3(d.K), b, 0, c->stream, d);
    if (LIN && d.coarse) {
        const size_t rows = (size_t)(d.tile_rows + d.max_halo);
        const size_t shm = sizeof(double) * 3 * rows + 3 * (rows + 8) + 16;
        const dim3 g(d.n_regblk);
        switch (d.T) {
            case 1: hipLaunchKernelGGL((k_coarse_tile<1>), g, b, shm, c->stream, d); break;
            case 2: hipLaunchKernelGGL((k_coarse_tile<2>), g, b, shm, c->stream, d); break;
            case 4: hipLaunchKernelGGL((k_coarse_tile<4>), g, b, shm, c->stream, d); break;
            case 16: hipLaunchKernelGGL((k_coarse_tile<16>), g, b, shm, c->stream, d); break;
            default: hipLaunchKernelGGL((k_coarse_tile<8>), g, b, shm, c->stream, d); break;
        }
        hipLaunchKernelGGL(k_coarse_reduce, dim3(1), b, 0, c->stream, d);
    }
    hipLaunchKernelGGL((k_finalize<LIN>), dim3(1), b, 0, c->stream, d);
    NRS_HIP(c, hipGetLastError());
    return NRS_OK;
}

static int read_scalars(nrs_ctx* c, Engine* e) {
    // k_finalize (always the last kernel enqueued before this) has written both mirrors
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    return NRS_OK;
}

// (H + lam I) x = b by block-Jacobi PCG, resumable: pcg_begin, then pcg_advance until it reports
// convergence; with stop_at_peek it also returns as soon as the 1e-4 milestone flag is up.
constexpr double PEEK_RTOL = 1e-1;      // inner-solve accuracy at which a trial is first evaluated
// Gain ratio below which a trial is rejected at the looks taken when the inner solve reaches 1e-2,
// 1e-3, 1e-4 (acceptance needs rho > 0).  The gain ratio of the partially converged step is within
// ~1e-2 / 4e-3 / 2e-3 of the final one at those milestones (second order in the PCG error; measured
// on the a2 and BA problems, profiles/README.md), so the thresholds keep a >10x margin.
constexpr int PEEK_LEVELS = 4;
constexpr double PEEK_MIN_REL_INCREASE = 1e-5;
constexpr double PEEK_RHO_LVL[5] = {0, -1.0, -0.25, -0.1, -0.03};

static int pcg_begin(nrs_ctx* c, Engine* e, double lam, int* it) {
    const Dev& d = e->d;
    NRS_HIP(c, hipMemsetAsync(d.flags, 0, sizeof(int) * 8, c->stream));
    if (d.coarse) hipLaunchKernelGGL(k_coarse_invert, dim3(1), dim3(BLK), sizeof(double) * (size_t)d.co_n * d.co_n, c->stream, d, lam);
    hipLaunchKernelGGL(k_trial_setup, dim3(d.n_vecblk), dim3(BLK), 0, c->stream, d, lam);
    *it = 0;
    return NRS_OK;
}

// enqueue one batch of PCG iterations (no host synchronisation)
static void pcg_enqueue_batch(nrs_ctx* c, Engine* e, double lam, int* it_io, int count = 0) {
    const Dev& d = e->d;
    const int n_poseblk = (d.K + 3) / 4;
    const double tol2 = c->opt.pcg_rtol * c->opt.pcg_rtol;
    int it = *it_io;
    const int stop = std::min(it + (count > 0 ? count : c->opt.pcg_batch), c->opt.pcg_max_iters);
    for (; it < stop; ++it) {
        if (d.fused) {
            const dim3 g(((d.n_regblk + 7) / 8) * 8), bb(BLK);
            const size_t shm = sizeof(double) * (6 * (size_t)(d.tile_rows + d.max_halo) + (d.coarse ? 9 * (size_t)d.n_regblk + 10 * CO_MAX : 0));
            switch (d.T) {
                case 1: if (d.coarse) hipLaunchKernelGGL((k_pcg_fused<1, true>), g, bb, shm, c->stream, d, lam, it, tol2, PEEK_RTOL * PEEK_RTOL);
                        else hipLaunchKernelGGL((k_pcg_fused<1, false>), g, bb, shm, c->stream, d, lam, it, tol2, PEEK_RTOL * PEEK_RTOL);
                        break;
                case 2: if (d.coarse) hipLaunchKernelGGL((k_pcg_fused<2, true>), g, bb, shm, c->stream, d, lam, it, tol2, PEEK_RTOL * PEEK_RTOL);
                        else hipLaunchKernelGGL((k_pcg_fused<2, false>), g, bb, shm, c->stream, d, lam, it, tol2, PEEK_RTOL * PEEK_RTOL);
                        break;
                case 4: if (d.coarse) hipLaunchKernelGGL((k_pcg_fused<4, true>), g, bb, shm, c->stream, d, lam, it, tol2, PEEK_RTOL * PEEK_RTOL);
                        else hipLaunchKernelGGL((k_pcg_fused<4, false>), g, bb, shm, c->stream, d, lam, it, tol2, PEEK_RTOL * PEEK_RTOL);
                        break;
                case 16: if (d.coarse) hipLaunchKernelGGL((k_pcg_fused<16, true>), g, bb, shm, c->stream, d, lam, it, tol2, PEEK_RTOL * PEEK_RTOL);
                        else hipLaunchKernelGGL((k_pcg_fused<16, false>), g, bb, shm, c->stream, d, lam, it, tol2, PEEK_RTOL * PEEK_RTOL);
                        break;
                default: if (d.coarse) hipLaunchKernelGGL((k_pcg_fused<8, true>), g, bb, shm, c->stream, d, lam, it, tol2, PEEK_RTOL * PEEK_RTOL);
                        else hipLaunchKernelGGL((k_pcg_fused<8, false>), g, bb, shm, c->stream, d, lam, it, tol2, PEEK_RTOL * PEEK_RTOL);
                        break;
            }
            continue;
        }
        {
            Timer t(c, &c->prof.spmv_ms, &c->prof.spmv_launches);
            launch_spmv(c, d, lam, it);
        }
        if (d.hier) hipLaunchKernelGGL(k_reduce_partials, dim3(1 + d.K), dim3(BLK), 0, c->stream, d);
        {
            Timer t(c, &c->prof.vec_ms, &c->prof.vec_launches);
            hipLaunchKernelGGL(k_pcg_update, dim3((((d.n_vecblk + 1) / 2 + 7) / 8) * 8 + n_poseblk), dim3(BLK), 0, c->stream,
                               d, lam, it, tol2, PEEK_RTOL * PEEK_RTOL);
        }
    }
    *it_io = it;
}

static int pcg_advance(nrs_ctx* c, Engine* e, double lam, int stop_level, int* it_io, bool* done) {
    const Dev& d = e->d;
    while (true) {
        // Once no peek is pending, batches are sized by the previous trial's iteration count (half of
        // what it predicts is left): fewer host round trips; launches past convergence are no-ops.
        int count = 0;
        if (stop_level == 0 && e->pred_iters > *it_io) count = std::min(std::max((e->pred_iters - *it_io) / 2, c->opt.pcg_batch), 8 * c->opt.pcg_batch);
        if (stop_level == 0 && e->pred_iters >= *it_io && e->pred_iters + 1 - *it_io <= c->opt.pcg_batch) count = e->pred_iters + 1 - *it_io;   // short solves: finish in one batch
        if (stop_level > 0 && e->pred_peek > 0) count = std::max(2, std::min(e->pred_peek, c->opt.pcg_batch));                                // waiting for a milestone: small steps
        pcg_enqueue_batch(c, e, lam, it_io, count);
        NRS_HIP(c, hipGetLastError());
        hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, c->stream, d);
        NRS_HIP(c, hipStreamSynchronize(c->stream));
        if (e->h_flags[0] || *it_io >= c->opt.pcg_max_iters) { *done = true; break; }
        if (stop_level && e->h_flags[3] >= stop_level) { *done = false; break; }
    }
    return NRS_OK;
}

// g2o SparseOptimizer::optimize + OptimizationAlgorithmLevenberg::solve on the resident problem.
// trace->count / iterations are accumulated (the drivers reset them once per entry point).
int engine_optimize(nrs_ctx* c, Engine* e, int iters, int round, nrs_lm_trace* trace) {
    if (iters < 0) return c->fail(NRS_ERR_INVALID, "iters < 0");
    NRS_HIP(c, hipSetDevice(c->device));
    Dev& d = e->d;
    double lam = -1, ni = 2;
    const bool peek_debug = getenv("NRS_PEEK_DEBUG") != nullptr;
    const bool check_chi = getenv("NRS_CHECK_CHI") != nullptr;
    double chi_carry = 0;
    const int peek_levels = peek_debug ? 4 : PEEK_LEVELS;
    for (int it = 0; it < iters; ++it) {
        NRS_TRY(evaluate<true>(c, e, e->cur));
        // computeActiveErrors at the start of an iteration re-derives the chi2 the last accepted trial
        // already produced (same state, same summation order): after the first iteration the host
        // does not wait for it, the linearisation and the first PCG batch go out back to back
        double chi = chi_carry;
        if (it == 0 || check_chi) {
            NRS_TRY(read_scalars(c, e));
            if (check_chi && it > 0 && e->h_scal[SC_CHI] != chi_carry)
                fprintf(stderr, "[nrs] chi2 carried %.17g vs recomputed %.17g\n", chi_carry, e->h_scal[SC_CHI]);
            chi = e->h_scal[SC_CHI];
        }
        if (it == 0) { lam = 1e-5 * e->h_scal[SC_MAXDIAG]; ni = 2; }
        if (!std::isfinite(chi) || !std::isfinite(lam)) return c->fail(NRS_ERR_NUMERIC, "non-finite chi2/lambda at LM iteration %d", it);
        double rho = 0;
        int qmax = 0;
        do {
            int pit = 0;
            bool done = false, early = false;
            const int trial = 1 - e->cur;
            double temp = 0, scale = 0;
            bool ok = true;
            NRS_TRY(pcg_begin(c, e, lam, &pit));
            auto eval_trial = [&]() -> int {
                Timer t(c, &c->prof.update_ms, &c->prof.update_launches);
                hipLaunchKernelGGL(k_apply, dim3(d.n_vecblk), dim3(BLK), 0, c->stream, d, lam, d.pose[e->cur], d.xl[e->cur], d.pose[trial], d.xl[trial]);
                NRS_TRY(evaluate<false>(c, e, trial));
                return read_scalars(c, e);                 // one synchronisation: chi2, scale and the PCG flags
            };
            const bool peeking = !c->opt.exact_trials;
            // The first batch of PCG iterations and a speculative evaluation of its result go out
            // together: most trials are decided by it (converged, or clearly rejected at a peek).  Its
            // size is what the previous trial needed to reach the first milestone (the kernels record
            // it), so a trial that is going to be rejected costs a handful of iterations.
            int seen = 0;                                  // peek levels already evaluated
            {
                int first = 0;
                // (after an iteration whose first trial was accepted, the next first trial usually is too:
                // short solves then go out whole, without the intermediate look)
                const bool expect_accept = qmax == 0 && e->first_trial_accepted && e->pred_iters > 0 && e->pred_iters + 1 <= 2 * c->opt.pcg_batch;
                if (peeking && !expect_accept) first = e->pred_peek > 0 ? std::min(e->pred_peek, c->opt.pcg_batch) : std::max(1, c->opt.pcg_batch / 2);
                else if (e->pred_iters > 0 && e->pred_iters + 1 <= 2 * c->opt.pcg_batch) first = e->pred_iters + 1;
                pcg_enqueue_batch(c, e, lam, &pit, first);
                NRS_TRY(eval_trial());
                done = e->h_flags[0] != 0 || pit >= c->opt.pcg_max_iters;
            }
            while (true) {
                temp = e->h_scal[SC_CHI];
                scale = e->h_scal[SC_SCALE] + 1e-3;
                if (done) break;
                const int lvl = e->h_flags[3];
                if (peeking && lvl > seen) {
                    // peek: a trial that is clearly going to be rejected is not solved any further --
                    // its step is discarded, so the iterate sequence is the reference's either way
                    const double rho_peek = (chi - temp) / scale;
                    // ... and only when the chi2 increase is well above the noise floor of the fp32
                    // projection (relative 1e-7 per evaluation): near convergence the gain ratio of a
                    // tiny step is noise over the 1e-3 regulariser of its denominator, at any accuracy
                    early = e->h_flags[2] == 0 && std::isfinite(temp) && rho_peek < PEEK_RHO_LVL[lvl] && (temp - chi) > PEEK_MIN_REL_INCREASE * chi;
                    if (peek_debug) { fprintf(stderr, "[peek] it %d trial %d lvl %d pit %d rho %.4f\n", it, qmax, lvl, pit, rho_peek); early = false; }
                    if (early) break;
                    seen = lvl;
                }
                NRS_TRY(pcg_advance(c, e, lam, peeking && seen < peek_levels ? seen + 1 : 0, &pit, &done));
                NRS_TRY(eval_trial());
            }
            ok = e->h_flags[2] == 0;
            if (!early && !ok) temp = 1.7976931348623157e308;
            if (!early) e->pred_iters = e->h_flags[1];
            if (e->h_flags[4] > 0) e->pred_peek = e->h_flags[4];
            if (qmax == 0) e->first_trial_accepted = !early && (chi - temp) / scale > 0 && std::isfinite(temp);
            const int inner = e->h_flags[1];
            rho = (chi - temp) / scale;
            if (peek_debug) fprintf(stderr, "[peek] it %d trial %d FINAL pit %d rho %.4f\n", it, qmax, pit, rho);
            const bool accepted = !early && rho > 0 && std::isfinite(temp);
            if (trace) {
                if (trace->trials && trace->count < trace->capacity) {
                    nrs_lm_trial& Tr = trace->trials[trace->count];
                    Tr.round = round; Tr.iter = it; Tr.trial = qmax; Tr.accepted = accepted; Tr.solver_ok = ok;
                    Tr.inner_iters = inner; Tr.early_rejected = early; Tr.reserved = 0;
                    Tr.lambda = lam; Tr.chi2 = chi; Tr.chi2_new = temp; Tr.rho = rho;
                }
                trace->count++;
            }
            if (accepted) {
                double alpha = 1.0 - std::pow(2 * rho - 1, 3);
                alpha = std::min(alpha, 2.0 / 3.0);
                lam *= std::max(1.0 / 3.0, alpha);
                ni = 2;
                chi = temp;
                e->cur = trial;                        // discardTop: the trial state becomes current
            } else {
                lam *= ni;
                ni *= 2;                               // pop: current state untouched
                if (!std::isfinite(lam)) break;
            }
            ++qmax;
        } while (rho < 0 && qmax < 10);
        chi_carry = chi;
        if (trace) trace->iterations++;
        if (qmax == 10 || rho == 0 || !std::isfinite(lam)) break;
    }
    return NRS_OK;
}

int engine_download(nrs_ctx* c, Engine* e, Pose* poses, double* x) {
    Dev& d = e->d;
    std::vector<double> xl((size_t)d.n_rows * 3);
    if (poses) NRS_HIP(c, hipMemcpyAsync(poses, d.pose[e->cur], sizeof(Pose) * d.K, hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipMemcpyAsync(xl.data(), d.xl[e->cur], sizeof(double) * xl.size(), hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    if (x)
        for (int v = 0; v < d.M; ++v)
            for (int k = 0; k < 3; ++k) x[3 * (size_t)v + k] = xl[3 * (size_t)e->vrow[v] + k];
    return NRS_OK;
}

int engine_residuals(nrs_ctx* c, Engine* e, double* r_reproj, double* r_spring, double* r_damper) {
    Dev& d = e->d;
    double* rr = e->t_out;
    double* rs = rr + 2 * (size_t)d.M;
    double* rd = rs + (size_t)d.n_sp;
    const int n = std::max(d.M, std::max(d.n_sp, d.n_dm));
    hipLaunchKernelGGL(k_tap_residuals, dim3((n + 255) / 256), dim3(256), 0, c->stream, d, d.pose[e->cur], d.xl[e->cur],
                       e->t_vrow, e->t_sp, e->t_d0, e->t_dm, e->t_w, rr, rs, rd);
    NRS_HIP(c, hipGetLastError());
    if (r_reproj) NRS_HIP(c, hipMemcpyAsync(r_reproj, rr, sizeof(double) * 2 * (size_t)d.M, hipMemcpyDeviceToHost, c->stream));
    if (r_spring && d.n_sp) NRS_HIP(c, hipMemcpyAsync(r_spring, rs, sizeof(double) * (size_t)d.n_sp, hipMemcpyDeviceToHost, c->stream));
    if (r_damper && d.n_dm) NRS_HIP(c, hipMemcpyAsync(r_damper, rd, sizeof(double) * 3 * (size_t)d.n_dm, hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    return NRS_OK;
}

int engine_edge_chi2(nrs_ctx* c, Engine* e, double* reproj, double* spring, double* damper) {
    Dev& d = e->d;
    std::vector<double> rr(2 * (size_t)d.M), rs((size_t)d.n_sp), rd(3 * (size_t)d.n_dm);
    NRS_TRY(engine_residuals(c, e, rr.data(), rs.data(), rd.data()));
    if (reproj)
        for (int i = 0; i < d.M; ++i) reproj[i] = d.info_reproj * (rr[2 * i] * rr[2 * i] + rr[2 * i + 1] * rr[2 * i + 1]);
    if (spring)
        for (int i = 0; i < d.n_sp; ++i) spring[i] = d.info_pos * rs[i] * rs[i];
    if (damper)
        for (int i = 0; i < d.n_dm; ++i)
            damper[i] = d.info_spatial * (rd[3 * i] * rd[3 * i] + rd[3 * i + 1] * rd[3 * i + 1] + rd[3 * i + 2] * rd[3 * i + 2]);
    return NRS_OK;
}

int engine_gradient(nrs_ctx* c, Engine* e, double* b, double* diag) {
    Dev& d = e->d;
    NRS_TRY(evaluate<true>(c, e, e->cur));
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
    for (int v = 0; v < d.M; ++v) {
        const size_t row = (size_t)e->vrow[v];
        for (int a = 0; a < 3; ++a) {
            b[6 * (size_t)d.K + 3 * (size_t)v + a] = bl[3 * row + a];
            diag[6 * (size_t)d.K + 3 * (size_t)v + a] = D[6 * row + d3[a]];
        }
    }
    return NRS_OK;
}

}  // namespace nrs
