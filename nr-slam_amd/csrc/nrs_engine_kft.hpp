// Keyframe-block direct solve of an EMBEDDED BA window (N2b, BASELINE configs[1] as written: points x graph nodes x keyframes).
// Part of nrs_engine.hip (one translation unit).
//
// What it replaces: g2o's exact linear solve per LM trial (reference third_party/g2o/g2o/solvers/eigen/linear_solver_eigen.h:92-173,
// block_solver.hpp:329-341) for LocalDeformableBundleAdjustment (modules/optimization/g2o_optimization.cc:880-1161) in the embedded form.
// The block-Jacobi PCG needs 110 .. 1000 iterations per LM trial on that window (2371 per optimize(5) at 5000 x 500 x 20): a keyframe's
// node copies are coupled densely through the shared skinned observations, and the depth components -- which a single view does not
// constrain -- are held by the temporal dampers only (profiles/r06_embedded_ba_precond_probe.txt: exact keyframe blocks alone still need
// 28 .. 306 iterations, per-node temporal blocks more than block-Jacobi).  The structure that IS exact:
//
//     H + lambda I  =  block tridiagonal over the keyframes,
//         diagonal block  A_k : the node copies of keyframe k and its pose -- dense (1.4k x 1.4k at 458 nodes): reprojection edges of the
//                               node copies, the skinned observations' hyper-edges (<= 11 node copies + the pose each), springs
//                               (OPT:1031-1072) and the same-keyframe halves of the dampers
//         coupling        T_k : keyframe k x keyframe k + 1 -- SPARSE: a damper (1c, 2c, 1n, 2n) (spatial_regularizer.cc:32-59, OPT:1076-1132)
//                               couples copies of two map points in consecutive keyframes, every block +- s I_3.
//
// Block elimination from BOTH ends of the window towards the middle keyframe m ("twisted"): two chains of K / 2 dependent steps
//     S_0 = A_0,  S_k = A_k - T_{k-1}^T G_{k-1} T_{k-1},   G_k = S_k^-1      (and mirrored from K - 1 down to m + 1; S_m takes both)
// with G_k formed EXPLICITLY by a blocked symmetric sweep (Gauss-Jordan on 64 x 64 pivot blocks: panel on the vector units, the rank-64
// trailing update on v_mfma_f64_16x16x4) -- so that a solve is a chain of dense matrix-vector products instead of triangular sweeps --
// and the Schur update is sparse x dense x sparse.  20 inversions of 1.4k x 1.4k = 53 GFLOP per LM trial.  The factorisation serves as
// the PRECONDITIONER of the engine's PCG (u = M^-1 r by one forward and one backward pass over the chains): M is H + lambda I up to the
// rounding of the explicit inverses, so the PCG converges in two or three iterations to the same pcg_rtol as before -- same LM iterates.
// Fixed-order sums throughout (no atomics): bit-reproducible.
#pragma once

namespace nrs {

constexpr int KFT_B = 64;            // pivot block / tile of the sweeps

struct KftDev {
    int K, ld, nb, m, nfm;           // keyframes; padded block dimension (multiple of KFT_B); ld / KFT_B; middle keyframe; ld / 3 (stride of the per-node tables)
    double* A;                       // K x ld x ld: A_k, then S_k, then G_k in place (full symmetric squares; rows / columns >= n_k: identity)
    double* YT;                      // 2 x ld x ld: per chain (G_f C^T)^T of the Schur update
    double* Bb; double* Cb;          // 2 (step parity) x 2 (chains) x nb x 64 x 64: panels of a sweep step, packed [k / 4][row][k % 4] (what the matrix-core operands read)
    double* Pv;                      // 2 (step parity) x 2 x 64 x 64: the swept pivot block (- P^-1)
    double* z; double* xs;           // K x ld: G_k y_k of the forward pass; the solution (compact order) for the backward pass
    double* vb;                      // 2 x ld: the right-hand side of a solve stage (per chain)
    const int* kf_nf; const int* kf_np;   // K: free node rows of a keyframe; 6 if its pose is free, else 0
    const int* kf_row;               // K x nfm: compact node -> row
    const int* row_ci;               // n_rows: row -> compact node of its keyframe (-1: fixed / padding)
    // diagonal-block assembly: the unique same-keyframe node pairs (hi > lo) with their contributions
    int n_pp;
    const uint32_t* pp_id;           // k << 24 | hi << 12 | lo
    const int* pp_ptr;               // n_pp + 1
    const uint32_t* pe_src;          // type << 30 | index: 0 spring (s_qc slot), 1 damper (d_s slot), 2 skinned observation (slot)
    const double* pe_w;              // skinned: om_hi om_lo
    // temporal couplings: unique pairs (a in keyframe k, b in keyframe k + 1), t = sum +- s
    int n_tp;
    const int* tp_ptr;               // n_tp + 1
    const uint32_t* te_src;          // sign << 31 | d_s slot
    double* tp_val;
    // coupling lists by the 'to' node, dir 0: from keyframe to - 1, dir 1: from keyframe to + 1
    const int* cl_ptr[2];            // K x (nfm + 1)
    const int* cl_from[2]; const int* cl_tp[2];
    double* cl_val[2];               // the lists' values (tp_val[cl_tp[.]], refreshed per linearisation: the solve's loops read them in place)
};

struct KftHost {
    bool on = false;
    KftDev d;
    size_t bytes = 0;
    int factorisations = 0;
    bool apply_pending = false;      // the PCG's iteration 0 ended with the residual test (k_kft_rnorm): u = M^-1 r is enqueued when the solve goes on
    std::vector<int> kf_nb;          // per keyframe: 64-blocks that hold unknowns (ceil((3 nodes + 6) / 64))
};

// ------------------------------------------------------------------------------------------------------------------ assembly
__global__ __launch_bounds__(256) void k_kft_clear(KftDev F) {
    const int k = blockIdx.y;
    const size_t idx = 2 * ((size_t)blockIdx.x * 256 + threadIdx.x), n2 = (size_t)F.ld * F.ld;
    if (idx >= n2) return;
    const int i = (int)(idx / F.ld), j = (int)(idx % F.ld), nk = 3 * F.kf_nf[k] + F.kf_np[k];
    double2 v = make_double2(0.0, 0.0);
    if (i >= nk && j == i) v.x = 1.0;
    if (i >= nk && j + 1 == i) v.y = 1.0;
    *reinterpret_cast<double2*>(F.A + (size_t)k * n2 + idx) = v;
}

// node diagonal blocks (D + lambda: every edge's share, the skinned observations' included) and the pose-node blocks: the node copy's own
// reprojection edge (factored form, as row_factored) + sum om B_o over the observations that reach the row (SK_RL lanes a row)
__global__ __launch_bounds__(BLK) void k_kft_diag(Dev P, KftDev F, double lam) {
    const int tid = threadIdx.x, r = blockIdx.x * SK_RPB + tid / SK_RL, t = tid % SK_RL;
    const int ci = F.row_ci[r], k = P.grp_pose[r / ROW_ALIGN];
    const int nf = F.kf_nf[k], np = F.kf_np[k];
    double acc[18];
#pragma unroll
    for (int q = 0; q < 18; ++q) acc[q] = 0;
    if (ci >= 0 && np) {
        for (int q = P.sk_row_q[2 * (size_t)r] + t; q < P.sk_row_q[2 * (size_t)r + 1]; q += SK_RL) {
            const double om = P.sk_rl_om[q];
            const double* rec = P.sk_rec + 27 * (size_t)P.sk_rl_obs[q] + 9;
#pragma unroll
            for (int j = 0; j < 18; ++j) acc[j] += om * rec[j];
        }
    }
#pragma unroll
    for (int q = 0; q < 18; ++q) acc[q] = sub_sum_t<SK_RL>(acc[q]);
    if (ci < 0 || t != 0) return;
    double* A = F.A + (size_t)k * F.ld * F.ld;
    const double* D = P.D + 6 * (size_t)r;
    const size_t o = (size_t)(3 * ci) * F.ld + 3 * ci;
    A[o] = D[0] + lam; A[o + 1] = D[1]; A[o + 2] = D[2];
    A[o + F.ld] = D[1]; A[o + F.ld + 1] = D[3] + lam; A[o + F.ld + 2] = D[4];
    A[o + 2 * (size_t)F.ld] = D[2]; A[o + 2 * (size_t)F.ld + 1] = D[4]; A[o + 2 * (size_t)F.ld + 2] = D[5] + lam;
    if (!np) return;
    const RowRec rc = P.rowrec[r];
    if (rc.w != 0.0) {
        const Pose Tcw = P.lin_pose[k];
        double R[9], xs[3];
        quat_to_R(Tcw.q, R);
#pragma unroll
        for (int a = 0; a < 3; ++a) xs[a] = P.lin_xl[3 * (size_t)r + a] + (P.X0 ? P.X0[3 * (size_t)r + a] : 0.0);
        const double px = R[0] * xs[0] + R[1] * xs[1] + R[2] * xs[2] + Tcw.t[0];
        const double py = R[3] * xs[0] + R[4] * xs[1] + R[5] * xs[2] + Tcw.t[1];
        const double pz = R[6] * xs[0] + R[7] * xs[1] + R[8] * xs[2] + Tcw.t[2];
        double Jl[2][3], Jp[2][6];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const double j0 = -(double)rc.J[3 * rr], j1 = -(double)rc.J[3 * rr + 1], j2 = -(double)rc.J[3 * rr + 2];
            Jp[rr][0] = -j1 * pz + j2 * py; Jp[rr][1] = j0 * pz - j2 * px; Jp[rr][2] = -j0 * py + j1 * px;
            Jp[rr][3] = j0; Jp[rr][4] = j1; Jp[rr][5] = j2;
            Jl[rr][0] = j0 * R[0] + j1 * R[3] + j2 * R[6];
            Jl[rr][1] = j0 * R[1] + j1 * R[4] + j2 * R[7];
            Jl[rr][2] = j0 * R[2] + j1 * R[5] + j2 * R[8];
        }
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) acc[3 * p + cc] += rc.w * (Jp[0][p] * Jl[0][cc] + Jp[1][p] * Jl[1][cc]);
    }
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            A[(size_t)(3 * nf + p) * F.ld + 3 * ci + cc] = acc[3 * p + cc];
            A[(size_t)(3 * ci + cc) * F.ld + 3 * nf + p] = acc[3 * p + cc];
        }
}

__global__ __launch_bounds__(256) void k_kft_pose(Dev P, KftDev F, double lam) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 36 * F.K) return;
    const int k = i / 36, p = (i % 36) / 6, q = i % 6;
    if (!F.kf_np[k]) return;
    const int lo = p < q ? p : q, hi = p < q ? q : p;
    const int pk = lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo);       // packed upper index (as k_pcg_update)
    const int nf = F.kf_nf[k];
    F.A[(size_t)k * F.ld * F.ld + (size_t)(3 * nf + p) * F.ld + 3 * nf + q] = P.Hpp[21 * k + pk] + (p == q ? lam : 0.0);
}

// KFT_PL lanes per unique same-keyframe node pair: a lane adds up every KFT_PL-th contribution of the pair's list in list order, the partial
// sums meet in a fixed butterfly (bit-reproducible).  One thread per pair walked lists of ~50 skinned observations with dependent fetches of
// their 27-double records: 183 us per LM trial at C2.
constexpr int KFT_PL = 8;
__global__ __launch_bounds__(256) void k_kft_pairs(Dev P, KftDev F) {
    const int t = blockIdx.x * 256 + threadIdx.x, i = t / KFT_PL, sub = t % KFT_PL;
    if (i >= F.n_pp) return;                                        // (whole groups of KFT_PL lanes leave together: the butterfly below stays inside a group)
    const uint32_t id = F.pp_id[i];
    const int k = (int)(id >> 24), hi = (int)((id >> 12) & 0xFFFu), lo = (int)(id & 0xFFFu);
    const size_t rh = (size_t)F.kf_row[k * F.nfm + hi], rl = (size_t)F.kf_row[k * F.nfm + lo];
    double v[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) v[a] = (P.lin_xl[3 * rh + a] + (P.X0 ? P.X0[3 * rh + a] : 0.0)) - (P.lin_xl[3 * rl + a] + (P.X0 ? P.X0[3 * rl + a] : 0.0));
    double b[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = F.pp_ptr[i] + sub; q < F.pp_ptr[i + 1]; q += KFT_PL) {
        const uint32_t src = F.pe_src[q];
        const uint32_t type = src >> 30, idx = src & 0x3FFFFFFFu;
        if (type == 0) {                                           // spring: - qc v v^T
            const double qc = P.s_qc[idx];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) b[3 * a + cc] -= qc * v[a] * v[cc];
        } else if (type == 1) {                                    // damper, the two vertices of one keyframe: - s I
            const double s = P.d_s[idx];
            b[0] -= s; b[4] -= s; b[8] -= s;
        } else {                                                   // skinned observation: om_hi om_lo J_l^T w J_l
            const double w = F.pe_w[q];
            const double* rec = P.sk_rec + 27 * (size_t)idx;
            b[0] += w * rec[0]; b[1] += w * rec[1]; b[2] += w * rec[2];
            b[3] += w * rec[1]; b[4] += w * rec[3]; b[5] += w * rec[4];
            b[6] += w * rec[2]; b[7] += w * rec[4]; b[8] += w * rec[5];
        }
    }
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int o = 1; o < KFT_PL; o <<= 1) b[a] += __shfl_xor(b[a], o, 64);
    if (sub) return;
    double* A = F.A + (size_t)k * F.ld * F.ld;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            A[(size_t)(3 * hi + a) * F.ld + 3 * lo + cc] = b[3 * a + cc];
            A[(size_t)(3 * lo + cc) * F.ld + 3 * hi + a] = b[3 * a + cc];
        }
}

__global__ __launch_bounds__(256) void k_kft_tvals(Dev P, KftDev F) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= F.n_tp) return;
    double t = 0;
    for (int q = F.tp_ptr[i]; q < F.tp_ptr[i + 1]; ++q) {
        const uint32_t src = F.te_src[q];
        const double s = P.d_s[src & 0x7FFFFFFFu];
        t += (src >> 31) ? -s : s;
    }
    F.tp_val[i] = t;
}

__global__ __launch_bounds__(256) void k_kft_clvals(KftDev F) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= F.n_tp) return;
    F.cl_val[0][i] = F.tp_val[F.cl_tp[0][i]];
    F.cl_val[1][i] = F.tp_val[F.cl_tp[1][i]];
}

// ------------------------------------------------------------------------------------------------------------------ the sweep
#include "nrs_kft_sweep.hpp"   // kft_sweep64_blk: the pivot block's sweep in 16-pivot steps (DPP broadcasts + matrix cores)

// SWEEP(j) of a symmetric matrix on 64 x 64 blocks (P = A_jj):  A_jj <- -P^-1,  A_Ij <- A_Ij P^-1 (and its mirror),
// A_IL <- A_IL - A_Ij P^-1 A_jL for I, L != j; after every block has been swept the matrix is -A^-1.  Two launches per step:
//   k_kft_panel  (one workgroup per block row I): inverts P in registers (kft_sweep64: sixteen 4-pivot steps),
//                B_I = C_I P^-1 on the vector units, leaves B_I in the panel (both triangles) and B_I / C_I in the packed buffers
//   k_kft_update (one workgroup per tile): A_IL -= B_I C_L^T on the matrix cores; the tile (j, j) takes the swept pivot block
// The final sign is taken off by the last step (k_kft_update with neg = 1 writes -A).  Both launches serve the two chains at once.
constexpr int KFT_LDP = KFT_B + 1;
constexpr size_t KFT_STEP_LDS = sizeof(double) * (2 * (size_t)(64 * 65) + 2 * 16 * 18 + 16 + 2 * (size_t)64 * 64) + 1024;   // the panel's buffers + two operand panels of 32 KB (~ 137 KB)
constexpr int KFT_CBS = 18;          // doubles per 4 x 4 block of the sweep's block-column buffer (16 + 2: sixteen lanes reading sixteen blocks hit sixteen bank groups)
constexpr size_t KFT_PANEL_LDS = sizeof(double) * (2 * (size_t)KFT_B * KFT_LDP + 2 * 16 * KFT_CBS + 16);   // P, C_I (padded rows), the sweep's block-column buffers + its -I block
// SWEEP of a 64 x 64 symmetric positive definite block held as 4 x 4 register blocks (thread (ti, tj) of 16 x 16 holds rows 4 ti .., columns
// 4 tj ..): sixteen steps of FOUR pivots each.  A step sweeps the 4 x 4 pivot block P = A_pp (every thread inverts it for itself -- a
// scalar sweep on sixteen registers), and with C_I = A_Ip (the block column, through LDS: one barrier per step instead of one per pivot)
//     A_IJ <- A_IJ - C_I P^-1 C_J^T,   A_Ip <- C_I P^-1,   A_pJ <- P^-1 C_J^T,   A_pp <- -P^-1
// as ONE expression  m a - L (P^-1 R^T):  L = C_I, R = C_J, m = 1 in general; in the pivot row L = -I (read from a constant block: the
// choice is an LDS ADDRESS, not sixteen selects), in the pivot column R = -I, and m = 0 in both.  345 cycles per pivot against 840 of the
// pivot-at-a-time form (one LDS round trip, one reciprocal chain and nine selects per pivot).  Leaves -A^-1 in a; colb: 2 x 16 x KFT_CBS + 16 doubles.
__device__ __forceinline__ bool kft_sweep64(double (&a)[4][4], double* colb, int ti, int tj) {
    double* negI = colb + 2 * 16 * KFT_CBS;
    if (ti == 0 && tj < 16) negI[tj] = (tj % 5 == 0) ? -1.0 : 0.0;   // (written before the first barrier below, read after it)
    bool bad = false;
#pragma unroll 1
    for (int pb = 0; pb < KFT_B / 4; ++pb) {
        double* cb = colb + (pb & 1) * 16 * KFT_CBS;
        if (tj == pb) {
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) cb[KFT_CBS * ti + 4 * x + y] = a[x][y];
        }
        __syncthreads();
        const bool prow = ti == pb, pcol = tj == pb;
        const double* Lp = prow ? negI : cb + KFT_CBS * ti;
        const double* Rp = pcol ? negI : cb + KFT_CBS * tj;
        double P[4][4], L[4][4];
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) { P[x][y] = cb[KFT_CBS * pb + 4 * x + y]; L[x][y] = Lp[4 * x + y]; }
        // - P^-1 by four scalar sweeps (lower triangle computed, mirrored)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double d = P[q][q];
            const bool ok = d > 0.0 && d < 1e300;
            bad = bad || !ok;
            d = ok ? d : 1.0;
            double inv = __builtin_amdgcn_rcp(d);
            inv = fma(fma(-d, inv, 1.0), inv, inv);
            double col[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) col[r] = P[r][q];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int cc = 0; cc <= r; ++cc) {
                    if (r == q || cc == q) continue;
                    P[r][cc] -= col[r] * col[cc] * inv;
                    P[cc][r] = P[r][cc];
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) if (r != q) { P[r][q] = col[r] * inv; P[q][r] = P[r][q]; }
            P[q][q] = -inv;
        }
        const double m = (prow || pcol) ? 0.0 : 1.0;
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            double R[4], T[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) R[t] = Rp[4 * y + t];
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) T[sidx] = -(P[sidx][0] * R[0] + P[sidx][1] * R[1] + P[sidx][2] * R[2] + P[sidx][3] * R[3]);   // (P holds -P^-1)
#pragma unroll
            for (int x = 0; x < 4; ++x) a[x][y] = m * a[x][y] - (L[x][0] * T[0] + L[x][1] * T[1] + L[x][2] * T[2] + L[x][3] * T[3]);
        }
    }
    return bad;
}

// SWEEP(j) of a symmetric matrix on 64 x 64 blocks (P = A_jj):  A_jj <- -P^-1,  A_Ij <- A_Ij P^-1 (and its mirror),
// A_IL <- A_IL - A_Ij P^-1 A_jL for I, L != j; after every block has been swept the matrix is -A^-1.  Two launches per step:
//   k_kft_panel  (one workgroup per block row I): inverts P in registers (kft_sweep64: sixteen 4-pivot steps),
//                B_I = C_I P^-1 on the vector units, leaves B_I in the panel (both triangles) and B_I / C_I in the packed buffers
//   k_kft_update (one workgroup per tile): A_IL -= B_I C_L^T on the matrix cores; the tile (j, j) takes the swept pivot block
// The final sign is taken off by the last step (k_kft_update with neg = 1 writes -A).  Both launches serve the two chains at once.

__global__ __launch_bounds__(256) void k_kft_panel(KftDev F, int j, int kf0, int kf1, int* flags) {
    extern __shared__ double sm[];
    const int ch = blockIdx.y, kf = ch ? kf1 : kf0;
    if (kf < 0) return;
    double* Ps = sm;
    double* Cs = sm + KFT_B * KFT_LDP;
    double* colb = sm + 2 * KFT_B * KFT_LDP;
    const int I = blockIdx.x, tid = threadIdx.x, ti = tid >> 4, tj = tid & 15, lane = tid & 63, w = tid >> 6;
    const int ld = F.ld;
    double* A = F.A + (size_t)kf * ld * ld;
    const double* Pt = A + (size_t)(KFT_B * j) * ld + KFT_B * j;
    const double* Ct = A + (size_t)(KFT_B * I) * ld + KFT_B * j;
    // both tiles through LDS: whole 512-byte rows per request (the register blocks and the matrix-core operands are read from there)
    for (int q = tid; q < KFT_B * KFT_B; q += 256) {
        const int r = q >> 6, cidx = q & 63;
        Ps[r * KFT_LDP + cidx] = Pt[(size_t)r * ld + cidx];
        Cs[r * KFT_LDP + cidx] = I != j ? Ct[(size_t)r * ld + cidx] : 0.0;
    }
    __syncthreads();
    double a[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) a[x][y] = Ps[(4 * ti + x) * KFT_LDP + 4 * tj + y];
    const bool bad = kft_sweep64(a, colb, ti, tj);
    if (bad && tid == 0) flags[2] = 1;
    const size_t tile = (size_t)KFT_B * KFT_B;
    double* Bb = F.Bb + ((size_t)ch * F.nb + I) * tile;
    double* Cb = F.Cb + ((size_t)ch * F.nb + I) * tile;
    if (I == j) {                                                   // the pivot block's own row: nothing to update with it; the swept block for k_kft_update
        double* Pv = F.Pv + (size_t)ch * tile;
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) Pv[(4 * ti + x) * KFT_B + 4 * tj + y] = a[x][y];
        for (int q = tid; q < KFT_B * KFT_B; q += 256) { Bb[q] = 0.0; Cb[q] = 0.0; }
        return;
    }
    __syncthreads();                                                // (every register block has been read from Ps)
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) Ps[(4 * ti + x) * KFT_LDP + 4 * tj + y] = -a[x][y];   // P^-1
    // C_I in the packed operand layout [k / 4][row][k % 4] (whole lines)
    for (int q = tid; q < KFT_B * KFT_B; q += 256) Cb[q] = Cs[((q >> 2) & 63) * KFT_LDP + 4 * (q >> 8) + (q & 3)];
    __syncthreads();
    // B_I = C_I P^-1 on the matrix cores: wave w its 16 rows, four 16-column tiles (P^-1 is symmetric: read by rows)
    nd_v4d c[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) c[n][g] = 0.0;
#pragma unroll 4
    for (int kq = 0; kq < KFT_B / 4; ++kq) {
        const double av = Cs[(16 * w + (lane & 15)) * KFT_LDP + 4 * kq + (lane >> 4)];
        double bv[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) bv[n] = Ps[(16 * n + (lane & 15)) * KFT_LDP + 4 * kq + (lane >> 4)];
#pragma unroll
        for (int n = 0; n < 4; ++n) c[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[n], c[n], 0, 0, 0);
    }
    // (wave w is the only reader of its rows of Cs: B_I takes their place)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) Cs[(16 * w + (lane >> 4) + 4 * g) * KFT_LDP + 16 * n + (lane & 15)] = c[n][g];
    __syncthreads();
    double* Bt = A + (size_t)(KFT_B * I) * ld + KFT_B * j;         // A_Ij <- B_I
    double* Btt = A + (size_t)(KFT_B * j) * ld + KFT_B * I;        // A_jI <- B_I^T
    for (int q = tid; q < KFT_B * KFT_B; q += 256) {
        const int r = q >> 6, cidx = q & 63;
        Bt[(size_t)r * ld + cidx] = Cs[r * KFT_LDP + cidx];
        Btt[(size_t)r * ld + cidx] = Cs[cidx * KFT_LDP + r];
        Bb[q] = Cs[((q >> 2) & 63) * KFT_LDP + 4 * (q >> 8) + (q & 3)];
    }
}

__global__ __launch_bounds__(256) void k_kft_update(KftDev F, int j, int kf0, int kf1, int neg) {
    const int ch = blockIdx.y, kf = ch ? kf1 : kf0;
    if (kf < 0) return;
    const int I = blockIdx.x / F.nb, L = blockIdx.x % F.nb;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ld = F.ld;
    const size_t tile = (size_t)KFT_B * KFT_B;
    double* At = F.A + (size_t)kf * ld * ld + (size_t)(KFT_B * I) * ld + KFT_B * L;
    const double sgn = neg ? -1.0 : 1.0;
    if (I == j && L == j) {
        const double* Pv = F.Pv + (size_t)ch * tile;
        for (int q = tid; q < KFT_B * KFT_B; q += 256) At[(size_t)(q / KFT_B) * ld + q % KFT_B] = sgn * Pv[q];
        return;
    }
    if (I == j || L == j) {
        if (neg) for (int q = tid; q < KFT_B * KFT_B; q += 256) { double* e = At + (size_t)(q / KFT_B) * ld + q % KFT_B; *e = -*e; }
        return;
    }
    const double* Bb = F.Bb + ((size_t)ch * F.nb + I) * tile;
    const double* Cb = F.Cb + ((size_t)ch * F.nb + L) * tile;
    nd_v4d c[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) c[n][g] = At[(size_t)(16 * w + (lane >> 4) + 4 * g) * ld + 16 * n + (lane & 15)];
#pragma unroll 4
    for (int kq = 0; kq < KFT_B / 4; ++kq) {
        const double av = -Bb[((size_t)kq * KFT_B + 16 * w + (lane & 15)) * 4 + (lane >> 4)];
        double bv[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) bv[n] = Cb[((size_t)kq * KFT_B + 16 * n + (lane & 15)) * 4 + (lane >> 4)];
#pragma unroll
        for (int n = 0; n < 4; ++n) c[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[n], c[n], 0, 0, 0);
    }
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) At[(size_t)(16 * w + (lane >> 4) + 4 * g) * ld + 16 * n + (lane & 15)] = sgn * c[n][g];
}

// ---- the two launches of a sweep step as ONE (look-ahead): launch j applies the trailing update of step j - 1 and, next to it, produces
// the panel of step j.  The panel workgroups (one per block row I: the critical path -- a 64-pivot sweep) bring the two tiles they need up
// to date themselves -- their own tile (I, j) and the pivot tile (j, j), each minus B C^T of step j - 1 on the matrix cores -- so they do
// not wait for the trailing update, which the other workgroups run meanwhile.  Nobody writes the pivot tile during the launch (every panel
// workgroup reads its pre-update value): the swept block travels through Pv and lands with launch j + 1.  Panels B / C / Pv are
// double-buffered by the parity of j.  nb + 1 launches per inversion instead of 2 nb; same arithmetic as the two-launch form up to the
// association of the pivot tile's update (NRS_KFT_TWO_LAUNCHES=1 selects that form).
// HW (with MF): EIGHT waves.  Waves 4 - 7 of a panel workgroup take the C_I half of the look-ahead -- this row's B panel, its tile, the rank-64
// update, C_I into LDS -- while waves 0 - 3 update and sweep the pivot block: that half (a panel through LDS, two barriers, sixty-four matrix
// instructions a wave) sat on the panel workgroup's path in front of the sweep.  The helpers meet the sweep's four barriers and the ones behind
// it, and share the stores; in a tile workgroup they leave at once.
template <bool MF, bool HW = false>                              // MF: the pivot block's sweep in 16-pivot steps on DPP broadcasts + the matrix cores (kft_sweep64_blk); else the 4-pivot register form
__global__ __launch_bounds__(HW ? 512 : 256) void k_kft_step(KftDev F, int j, int kf0, int kf1, int* flags, int nbu) {   // nbu: blocks in use (the larger of the two keyframes': rows beyond a keyframe's unknowns are identity rows, whole identity blocks need no sweep)
    extern __shared__ double sm[];
    // (1-D grid: the panel workgroups of BOTH chains come first -- with the chain as the grid's second dimension chain 1's panel workgroups were
    // dispatched behind chain 0's 500 tiles, a round of tiles late: 40 against 34 us a launch when two chains are in flight)
    const int ch = (int)blockIdx.x < 2 * nbu ? (int)blockIdx.x / nbu : ((int)blockIdx.x - 2 * nbu) / (nbu * nbu);
    const int bx = (int)blockIdx.x < 2 * nbu ? (int)blockIdx.x % nbu : nbu + ((int)blockIdx.x - 2 * nbu) % (nbu * nbu);
    const int kf = ch ? kf1 : kf0;
    if (kf < 0) return;
    const int nb = F.nb, ld = F.ld, jp = j - 1;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const size_t tile = (size_t)KFT_B * KFT_B, pbuf = (size_t)2 * nb * tile;
    double* A = F.A + (size_t)kf * ld * ld;
    const double* Bp = F.Bb + (size_t)(jp & 1) * pbuf + (size_t)ch * nb * tile;       // panels of step j - 1
    const double* Cp = F.Cb + (size_t)(jp & 1) * pbuf + (size_t)ch * nb * tile;
    const double* Pvp = F.Pv + ((size_t)(jp & 1) * 2 + ch) * tile;
    const bool last = j == nbu;                                     // the closing launch: trailing update of the last step, the sign taken off
    const double sgn = last ? -1.0 : 1.0;
    // tile (I, L) <- sgn * (tile - B_I C_L^T) on the matrix cores, to global memory or (row-major, stride KFT_LDP) to LDS
    auto updated_tile = [&](int I, int L, bool apply, double* lds_out) {
        const double* At = A + (size_t)(KFT_B * I) * ld + KFT_B * L;
        nd_v4d c[4];
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) c[n][g] = At[(size_t)(16 * w + (lane >> 4) + 4 * g) * ld + 16 * n + (lane & 15)];
        if (apply) {
            const double* Bb = Bp + (size_t)I * tile;
            const double* Cb = Cp + (size_t)L * tile;
#pragma unroll 4
            for (int kq = 0; kq < KFT_B / 4; ++kq) {
                const double av = -Bb[((size_t)kq * KFT_B + 16 * w + (lane & 15)) * 4 + (lane >> 4)];
                double bv[4];
#pragma unroll
                for (int n = 0; n < 4; ++n) bv[n] = Cb[((size_t)kq * KFT_B + 16 * n + (lane & 15)) * 4 + (lane >> 4)];
#pragma unroll
                for (int n = 0; n < 4; ++n) c[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[n], c[n], 0, 0, 0);
            }
        }
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = 16 * w + (lane >> 4) + 4 * g, cc = 16 * n + (lane & 15);
                if (lds_out) lds_out[r * KFT_LDP + cc] = c[n][g];
                else A[(size_t)(KFT_B * I + r) * ld + KFT_B * L + cc] = sgn * c[n][g];
            }
    };
    static_assert(MF || !HW, "the helper waves go with the 16-pivot sweep");
    if (bx >= nbu) {
        // ---- trailing update of step j - 1 (every tile outside its pivot row / column and outside this step's panel)
        if (jp < 0) return;
        if (HW && tid >= 256) return;                               // (a tile is four waves' work)
        const int q = bx - nbu, I = q / nbu, L = q % nbu;
        double* At = A + (size_t)(KFT_B * I) * ld + KFT_B * L;
        if (I == jp && L == jp) {                                   // the swept pivot block of step j - 1 lands
            for (int e = tid; e < KFT_B * KFT_B; e += 256) At[(size_t)(e / KFT_B) * ld + e % KFT_B] = sgn * Pvp[e];
            return;
        }
        if (I == jp || L == jp) {                                   // its panel: B, already in place
            if (last) for (int e = tid; e < KFT_B * KFT_B; e += 256) { double* x = At + (size_t)(e / KFT_B) * ld + e % KFT_B; *x = -*x; }
            return;
        }
        if (!last && (I == j || L == j)) return;                    // this step's panel and pivot tile: the panel workgroups'
        if (I < L) return;                                          // the matrix is symmetric: the lower tiles are computed, the upper ones are their mirrors
        if (I == L) { updated_tile(I, L, true, nullptr); return; }
        double* Ts = sm;                                            // (row-major, stride KFT_LDP: the transposed copy is read from here, whole rows per store)
        updated_tile(I, L, true, Ts);
        __syncthreads();
        double* Al = A + (size_t)(KFT_B * I) * ld + KFT_B * L;
        double* Au = A + (size_t)(KFT_B * L) * ld + KFT_B * I;
        for (int e = tid; e < KFT_B * KFT_B; e += 256) {
            const int r = e >> 6, cidx = e & 63;
            Al[(size_t)r * ld + cidx] = sgn * Ts[r * KFT_LDP + cidx];
            Au[(size_t)r * ld + cidx] = sgn * Ts[cidx * KFT_LDP + r];
        }
        return;
    }
    if (last) return;
    // ---- panel of step j: block row I
    const int I = bx, ti = tid >> 4, tj = tid & 15;
    double* Ps = sm;
    double* Cs = sm + KFT_B * KFT_LDP;
    double* colb = sm + 2 * KFT_B * KFT_LDP;
    // The two tiles this workgroup needs, brought up to date with step j - 1: P = tile (j, j) and C_I = tile (I, j), each minus B C^T.  The three
    // operand panels (B_j, C_j, B_I of step j - 1: 32 KB each) come through LDS in whole lines -- the matrix cores read them from there --
    // instead of 8-byte global loads per operand (9 us of this workgroup's ~35 before).
    double* X0 = sm + 2 * KFT_B * KFT_LDP + 2 * 16 * KFT_CBS + 16;  // B panel (packed [k / 4][row][k % 4])
    double* X1 = X0 + KFT_B * KFT_B;                                // C_j
    const bool upP = jp >= 0, upC = jp >= 0 && I != j && I != jp;  // (the previous pivot row holds B already: no update)
    // EVERY global request of the look-ahead goes out before the first wait -- the pivot row's two operand panels, this row's B panel (held in
    // registers until X0 is free again) and both tiles: as four dependent phases (panels, pivot tile, B_I, own tile) the launch paid four
    // memory round trips under the load of the trailing update's workgroups, ~10 of its 34 us
    const double* Pt = A + (size_t)(KFT_B * j) * ld + KFT_B * j;
    const double* Ct = A + (size_t)(KFT_B * I) * ld + KFT_B * j;
    const bool helper = HW && tid >= 256;                           // (HW: waves 4 - 7)
    const int ht = HW ? (tid & 255) : tid, wr = HW ? (w & 3) : w;   // a thread's place among its four waves; its wave's sixteen rows
    constexpr int NT = HW ? 512 : 256;
    double rbj[16], rcj[16], rbi[16];                               // (plain doubles: arrays of 16-byte vectors filled in unrolled loops end up in scratch memory)
    nd_v4d cp[4], cc[4];
    if (upP && !helper) {
        const double2* b2 = reinterpret_cast<const double2*>(Bp + (size_t)j * tile);
        const double2* c2 = reinterpret_cast<const double2*>(Cp + (size_t)j * tile);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const double2 vb = b2[ht + 256 * q], vc = c2[ht + 256 * q];
            rbj[2 * q] = vb.x; rbj[2 * q + 1] = vb.y; rcj[2 * q] = vc.x; rcj[2 * q + 1] = vc.y;
        }
    }
    if (upC && (helper || !HW)) {
        const double2* b2 = reinterpret_cast<const double2*>(Bp + (size_t)I * tile);
#pragma unroll
        for (int q = 0; q < 8; ++q) { const double2 vb = b2[ht + 256 * q]; rbi[2 * q] = vb.x; rbi[2 * q + 1] = vb.y; }
    }
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const size_t o = (size_t)(16 * wr + (lane >> 4) + 4 * g) * ld + 16 * n + (lane & 15);
            cp[n][g] = !helper ? Pt[o] : 0.0;
            cc[n][g] = I != j && (helper || !HW) ? Ct[o] : 0.0;
        }
    auto rank64 = [&](nd_v4d (&c)[4], const double* Xl) {           // c -= Xl X1^T on the matrix cores
#pragma unroll 4
        for (int kq = 0; kq < KFT_B / 4; ++kq) {
            const double av = -Xl[(kq * KFT_B + 16 * wr + (lane & 15)) * 4 + (lane >> 4)];
            double bv[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) bv[n] = X1[(kq * KFT_B + 16 * n + (lane & 15)) * 4 + (lane >> 4)];
#pragma unroll
            for (int n = 0; n < 4; ++n) c[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[n], c[n], 0, 0, 0);
        }
    };
    auto to_lds = [&](const nd_v4d (&c)[4], double* out) {          // (row-major, stride KFT_LDP)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) out[(16 * wr + (lane >> 4) + 4 * g) * KFT_LDP + 16 * n + (lane & 15)] = c[n][g];
    };
    if constexpr (HW) {
        if (!helper) {
            if (upP) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    reinterpret_cast<double2*>(X0)[ht + 256 * q] = make_double2(rbj[2 * q], rbj[2 * q + 1]);
                    reinterpret_cast<double2*>(X1)[ht + 256 * q] = make_double2(rcj[2 * q], rcj[2 * q + 1]);
                }
            }
        } else if (upC) {                                           // (P's area is free until the sweep is over: this row's B panel)
#pragma unroll
            for (int q = 0; q < 8; ++q) reinterpret_cast<double2*>(Ps)[ht + 256 * q] = make_double2(rbi[2 * q], rbi[2 * q + 1]);
        }
        __syncthreads();
        if (!helper) { if (upP) rank64(cp, X0); }
        else {
            if (upC) rank64(cc, Ps);
            if (I != j) to_lds(cc, Cs);
        }
        __syncthreads();                                            // (X0 has been read: the sweep's scratch; C_I is in place)
    } else {
        if (upP) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                reinterpret_cast<double2*>(X0)[tid + 256 * q] = make_double2(rbj[2 * q], rbj[2 * q + 1]);
                reinterpret_cast<double2*>(X1)[tid + 256 * q] = make_double2(rcj[2 * q], rcj[2 * q + 1]);
            }
            __syncthreads();
            rank64(cp, X0);
        }
        if (!MF) to_lds(cp, Ps);                                    // (the 16-pivot sweep takes the block from the registers)
        if (I != j) {
            if (upC) {
                __syncthreads();                                    // (every wave has read B_j from X0)
#pragma unroll
                for (int q = 0; q < 8; ++q) reinterpret_cast<double2*>(X0)[tid + 256 * q] = make_double2(rbi[2 * q], rbi[2 * q + 1]);
                __syncthreads();
                rank64(cc, X0);
            }
            to_lds(cc, Cs);
        }
        __syncthreads();
    }
    double* Bb = F.Bb + (size_t)(j & 1) * pbuf + ((size_t)ch * nb + I) * tile;
    double* Cb = F.Cb + (size_t)(j & 1) * pbuf + ((size_t)ch * nb + I) * tile;
    if constexpr (MF) {
        KftTiles pc;
        pc.t0 = cp[0]; pc.t1 = cp[1]; pc.t2 = cp[2]; pc.t3 = cp[3];
        if (!helper) {
            const bool bad = kft_sweep64_blk(pc, X0, lane, w);       // (the operand panels are not needed any more)
            if (bad && lane == 0) flags[2] = 1;
        } else {
            if (I != j)                                             // (C_I's packed copy for the next step's tiles goes out while the others sweep)
                for (int q = ht; q < KFT_B * KFT_B; q += 256) Cb[q] = Cs[((q >> 2) & 63) * KFT_LDP + 4 * (q >> 8) + (q & 3)];
#pragma unroll
            for (int q = 0; q < 4; ++q) __syncthreads();            // (the sweep's four: one per 16-pivot step)
        }
        if (I == j) {
            if (!helper) {
                double* Pv = F.Pv + ((size_t)(j & 1) * 2 + ch) * tile;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    double* pw = Pv + (16 * w + (lane >> 4) + 4 * g) * KFT_B + (lane & 15);
                    pw[0] = pc.t0[g]; pw[16] = pc.t1[g]; pw[32] = pc.t2[g]; pw[48] = pc.t3[g];
                }
            }
            return;                                                 // (its own B / C slots are never read: the trailing update skips the pivot row / column)
        }
        __syncthreads();
        if (!helper) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {                           // P^-1
                double* pw = Ps + (16 * w + (lane >> 4) + 4 * g) * KFT_LDP + (lane & 15);
                pw[0] = -pc.t0[g]; pw[16] = -pc.t1[g]; pw[32] = -pc.t2[g]; pw[48] = -pc.t3[g];
            }
        }
    } else {
    double a[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) a[x][y] = Ps[(4 * ti + x) * KFT_LDP + 4 * tj + y];
    const bool bad = kft_sweep64(a, colb, ti, tj);
    if (bad && tid == 0) flags[2] = 1;
    if (I == j) {
        double* Pv = F.Pv + ((size_t)(j & 1) * 2 + ch) * tile;
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) Pv[(4 * ti + x) * KFT_B + 4 * tj + y] = a[x][y];
        return;                                                     // (its own B / C slots are never read: the trailing update skips the pivot row / column)
    }
    __syncthreads();
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) Ps[(4 * ti + x) * KFT_LDP + 4 * tj + y] = -a[x][y];   // P^-1
    }
    if (!HW)
        for (int q = tid; q < KFT_B * KFT_B; q += NT) Cb[q] = Cs[((q >> 2) & 63) * KFT_LDP + 4 * (q >> 8) + (q & 3)];
    __syncthreads();
    if (!helper) {
        nd_v4d c[4];
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) c[n][g] = 0.0;
#pragma unroll 4
        for (int kq = 0; kq < KFT_B / 4; ++kq) {
            const double av = Cs[(16 * w + (lane & 15)) * KFT_LDP + 4 * kq + (lane >> 4)];
            double bv[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) bv[n] = Ps[(16 * n + (lane & 15)) * KFT_LDP + 4 * kq + (lane >> 4)];
#pragma unroll
            for (int n = 0; n < 4; ++n) c[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[n], c[n], 0, 0, 0);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) Cs[(16 * w + (lane >> 4) + 4 * g) * KFT_LDP + 16 * n + (lane & 15)] = c[n][g];   // (a wave reads and writes its own sixteen rows)
    }
    __syncthreads();
    double* Bt = A + (size_t)(KFT_B * I) * ld + KFT_B * j;
    double* Btt = A + (size_t)(KFT_B * j) * ld + KFT_B * I;
    for (int q = tid; q < KFT_B * KFT_B; q += NT) {
        const int r = q >> 6, cidx = q & 63;
        Bt[(size_t)r * ld + cidx] = Cs[r * KFT_LDP + cidx];
        Btt[(size_t)r * ld + cidx] = Cs[cidx * KFT_LDP + r];
        Bb[q] = Cs[((q >> 2) & 63) * KFT_LDP + 4 * (q >> 8) + (q & 3)];
    }
}

// ------------------------------------------------------------------------------------------------------------------ Schur update
// chain 0: from keyframe f0 to f0 + 1 (dir 0); chain 1: from f1 to f1 - 1 (dir 1).  C = the coupling block (rows: 'to' nodes, columns: 'from'
// nodes), S_to -= C G_from C^T in two passes through YT = (G_from C^T)^T (G is symmetric: every access runs along a row)
__global__ __launch_bounds__(256) void k_kft_gct(KftDev F, int f0, int f1) {
    const int ch = blockIdx.z, f = ch ? f1 : f0;
    if (f < 0) return;
    const int t = ch ? f - 1 : f + 1, dir = ch;
    // a thread takes the three components of a 'to' node at once: the launch is bound by the dependent chain list entry -> row of G, not by
    // bytes -- one column (node, component) a workgroup: 43 us; three accumulators a thread and two list entries in flight: a third of the workgroups
    const int b = blockIdx.y;
    if (b >= F.kf_nf[t]) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= F.ld) return;
    const int ld = F.ld;
    const double* G = F.A + (size_t)f * ld * ld + i;
    const int* ptr = F.cl_ptr[dir] + (size_t)t * (F.nfm + 1);
    const int e0 = ptr[b], e1 = ptr[b + 1];
    double a0 = 0, a1 = 0, a2 = 0;
    int e = e0;
    for (; e + 1 < e1; e += 2) {                                    // (sums in list order)
        const double v0 = F.cl_val[dir][e], v1 = F.cl_val[dir][e + 1];
        const double* g0 = G + (size_t)(3 * F.cl_from[dir][e]) * ld;
        const double* g1 = G + (size_t)(3 * F.cl_from[dir][e + 1]) * ld;
        const double x0 = g0[0], x1 = g0[ld], x2 = g0[2 * (size_t)ld], y0 = g1[0], y1 = g1[ld], y2 = g1[2 * (size_t)ld];
        a0 += v0 * x0; a1 += v0 * x1; a2 += v0 * x2;
        a0 += v1 * y0; a1 += v1 * y1; a2 += v1 * y2;
    }
    if (e < e1) {
        const double v0 = F.cl_val[dir][e];
        const double* g0 = G + (size_t)(3 * F.cl_from[dir][e]) * ld;
        a0 += v0 * g0[0]; a1 += v0 * g0[ld]; a2 += v0 * g0[2 * (size_t)ld];
    }
    double* Y = F.YT + ((size_t)ch * ld + 3 * b) * ld + i;
    Y[0] = a0; Y[ld] = a1; Y[2 * (size_t)ld] = a2;
}
// one workgroup per KFT_TC columns of S_to: their YT rows in LDS, a thread per 'to' node -- its list is read once for all the columns
// and its three components
constexpr int KFT_TC = 4;
constexpr int KFT_TGT_NT = 512;      // (a keyframe's ~460 nodes in one pass)
__global__ __launch_bounds__(KFT_TGT_NT) void k_kft_tgt(KftDev F, int f0, int f1) {
    extern __shared__ double yrow[];
    const int ch = blockIdx.y, f = ch ? f1 : f0;
    if (f < 0) return;
    const int t = ch ? f - 1 : f + 1, dir = ch;
    const int col0 = blockIdx.x * KFT_TC;
    const int nft = F.kf_nf[t], ld = F.ld;
    if (col0 >= 3 * nft) return;
    const int nc = min(KFT_TC, 3 * nft - col0);
    const double* Y = F.YT + ((size_t)ch * ld + col0) * ld;
    for (int i = threadIdx.x; i < nc * ld; i += KFT_TGT_NT) yrow[i] = Y[i];
    __syncthreads();
    const int* ptr = F.cl_ptr[dir] + (size_t)t * (F.nfm + 1);
    double* S = F.A + (size_t)t * ld * ld + (size_t)col0 * ld;
    for (int b = threadIdx.x; b < nft; b += KFT_TGT_NT) {
        double acc[KFT_TC][3];
#pragma unroll
        for (int q = 0; q < KFT_TC; ++q) acc[q][0] = acc[q][1] = acc[q][2] = 0.0;
        const int e0 = ptr[b], e1 = ptr[b + 1];
        int e = e0;
        for (; e + 1 < e1; e += 2) {                                // (two list entries in flight; sums in list order)
            const double v = F.cl_val[dir][e], v2 = F.cl_val[dir][e + 1];
            const int o = 3 * F.cl_from[dir][e], o2 = 3 * F.cl_from[dir][e + 1];
#pragma unroll
            for (int q = 0; q < KFT_TC; ++q) {
                const double* y = yrow + q * ld + o;
                const double* y2 = yrow + q * ld + o2;
                acc[q][0] += v * y[0]; acc[q][1] += v * y[1]; acc[q][2] += v * y[2];
                acc[q][0] += v2 * y2[0]; acc[q][1] += v2 * y2[1]; acc[q][2] += v2 * y2[2];
            }
        }
        if (e < e1) {
            const double v = F.cl_val[dir][e];
            const int o = 3 * F.cl_from[dir][e];
#pragma unroll
            for (int q = 0; q < KFT_TC; ++q) {
                const double* y = yrow + q * ld + o;
                acc[q][0] += v * y[0]; acc[q][1] += v * y[1]; acc[q][2] += v * y[2];
            }
        }
#pragma unroll
        for (int q = 0; q < KFT_TC; ++q)
            if (q < nc) { double* Sq = S + (size_t)q * ld + 3 * b; Sq[0] -= acc[q][0]; Sq[1] -= acc[q][1]; Sq[2] -= acc[q][2]; }
    }
}

// ------------------------------------------------------------------------------------------------------------------ the solve
// One stage = one keyframe per chain, two launches: k_kft_vec builds the right-hand side (the residual of the keyframe's unknowns and the
// sparse coupling with the neighbour's vector, one thread per unknown), k_kft_gemv multiplies by G_k (16 rows per workgroup).
//   mode 0: forward   z_k = G_k (r_k - C z_prev)
//   mode 1: middle    x_m = G_m (r_m - C z_{m-1} - C' z_{m+1})
//   mode 2: backward  x_k = z_k - G_k C x_next
// x is scattered to the PCG's u (rows / poses).
__global__ __launch_bounds__(256) void k_kft_vec(KftDev F, int mode, int k0, int k1, const double* __restrict__ rv, const double* __restrict__ rp,
                                                 const int* __restrict__ flags) {
    const int ch = blockIdx.y, k = ch ? k1 : k0;
    if (k < 0 || flags[0]) return;                                  // (launches queued behind a converged solve are no-ops, as the PCG's own)
    const int jx = blockIdx.x * 256 + threadIdx.x;
    if (jx >= F.ld) return;
    const int ld = F.ld, nf = F.kf_nf[k], np = F.kf_np[k], K = F.K;
    // the neighbours whose vectors enter: forward -- the previous keyframe of the chain; backward -- the next towards the middle
    int nbr[2] = {-1, -1};
    if (mode == 0) nbr[0] = ch ? (k + 1 < K ? k + 1 : -1) : k - 1;
    else if (mode == 2) nbr[0] = ch ? k - 1 : k + 1;
    else { nbr[0] = k - 1; nbr[1] = k + 1 < K ? k + 1 : -1; }
    const double* src = mode == 2 ? F.xs : F.z;
    double v = 0;
    if (jx < 3 * nf) {
        const int b = jx / 3, comp = jx % 3;
        if (mode != 2) v = rv[3 * (size_t)F.kf_row[k * F.nfm + b] + comp];
        double s = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int o = nbr[h];
            if (o < 0) continue;
            const int dir = o < k ? 0 : 1;
            const int* ptr = F.cl_ptr[dir] + (size_t)k * (F.nfm + 1);
            const double* ov = src + (size_t)o * ld;
            for (int e = ptr[b]; e < ptr[b + 1]; ++e) s += F.cl_val[dir][e] * ov[3 * F.cl_from[dir][e] + comp];
        }
        v = mode == 2 ? s : v - s;
    } else if (jx < 3 * nf + np) {
        if (mode != 2) v = rp[6 * k + (jx - 3 * nf)];
    }
    F.vb[(size_t)ch * ld + jx] = v;
}

__global__ __launch_bounds__(256) void k_kft_gemv(KftDev F, int mode, int k0, int k1, double* __restrict__ uv, double* __restrict__ up, const int* __restrict__ flags) {
    extern __shared__ double vec[];
    const int ch = blockIdx.y, k = ch ? k1 : k0;
    if (k < 0 || flags[0]) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ld = F.ld, nf = F.kf_nf[k], np = F.kf_np[k];
    for (int jx = tid; jx < ld; jx += 256) vec[jx] = F.vb[(size_t)ch * ld + jx];
    const double* G = F.A + (size_t)k * ld * ld;
    const int row0 = blockIdx.x * 16 + 4 * w;
    // (the rows of G are requested before the barrier: the two are independent)
    double acc[4] = {0, 0, 0, 0};
    __syncthreads();
    for (int c0 = 2 * lane; c0 < ld; c0 += 128) {
        const double v0 = vec[c0], v1 = vec[c0 + 1];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const double2 g = *reinterpret_cast<const double2*>(G + (size_t)(row0 + a) * ld + c0);
            acc[a] += g.x * v0 + g.y * v1;
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a] = wave_sum(acc[a]);
    if (lane < 4) {
        const int row = row0 + lane;
        double x = acc[0];
#pragma unroll
        for (int a = 1; a < 4; ++a) x = lane == a ? acc[a] : x;
        if (mode == 0) { F.z[(size_t)k * ld + row] = x; return; }
        if (mode == 2) x = F.z[(size_t)k * ld + row] - x;
        F.xs[(size_t)k * ld + row] = x;
        if (row < 3 * nf) uv[3 * (size_t)F.kf_row[k * F.nfm + row / 3] + row % 3] = x;
        else if (row < 3 * nf + np) up[6 * k + (row - 3 * nf)] = x;
    }
}

// ------------------------------------------------------------------------------------------------------------------ host
static int kft_invert(nrs_ctx* c, const KftHost& H, int kf0, int kf1, int* flags) {
    const KftDev& F = H.d;
    if (c->env("NRS_KFT_TWO_LAUNCHES")) {                           // (A/B: the panel and the trailing update of a step as launches of their own)
        for (int j = 0; j < F.nb; ++j) {
            hipLaunchKernelGGL(k_kft_panel, dim3(F.nb, 2), dim3(256), KFT_PANEL_LDS, c->stream, F, j, kf0, kf1, flags);
            hipLaunchKernelGGL(k_kft_update, dim3(F.nb * F.nb, 2), dim3(256), 0, c->stream, F, j, kf0, kf1, j + 1 == F.nb ? 1 : 0);
        }
        return NRS_OK;
    }
    const int nbu = std::max(kf0 >= 0 ? H.kf_nb[kf0] : 1, kf1 >= 0 ? H.kf_nb[kf1] : 1);
    for (int j = 0; j <= nbu; ++j)
        if (c->env("NRS_KFT_SCALAR_SWEEP")) hipLaunchKernelGGL(k_kft_step<false>, dim3(2 * (nbu + nbu * nbu)), dim3(256), KFT_STEP_LDS, c->stream, F, j, kf0, kf1, flags, nbu);
        else if (c->env("NRS_KFT_FOUR_WAVES")) hipLaunchKernelGGL(k_kft_step<true>, dim3(2 * (nbu + nbu * nbu)), dim3(256), KFT_STEP_LDS, c->stream, F, j, kf0, kf1, flags, nbu);
        else hipLaunchKernelGGL((k_kft_step<true, true>), dim3(2 * (nbu + nbu * nbu)), dim3(512), KFT_STEP_LDS, c->stream, F, j, kf0, kf1, flags, nbu);
    return NRS_OK;
}

// assembly of every A_k and T_k at the current linearisation (after evaluate<true>), then the two elimination chains
static int kft_factor(nrs_ctx* c, Engine* e, KftHost* H, double lam) {
    const Dev& d = e->d;
    const KftDev& F = H->d;
    const size_t n2 = (size_t)F.ld * F.ld;
    hipLaunchKernelGGL(k_kft_clear, dim3((unsigned)((n2 / 2 + 255) / 256), F.K), dim3(256), 0, c->stream, F);
    hipLaunchKernelGGL(k_kft_diag, dim3(d.n_rows / SK_RPB), dim3(BLK), 0, c->stream, d, F, lam);
    hipLaunchKernelGGL(k_kft_pose, dim3((36 * F.K + 255) / 256), dim3(256), 0, c->stream, d, F, lam);
    if (F.n_pp) hipLaunchKernelGGL(k_kft_pairs, dim3((unsigned)(((size_t)F.n_pp * KFT_PL + 255) / 256)), dim3(256), 0, c->stream, d, F);
    if (F.n_tp) {
        hipLaunchKernelGGL(k_kft_tvals, dim3((F.n_tp + 255) / 256), dim3(256), 0, c->stream, d, F);
        hipLaunchKernelGGL(k_kft_clvals, dim3((F.n_tp + 255) / 256), dim3(256), 0, c->stream, F);
    }
    const int len0 = F.m, len1 = F.K - 1 - F.m;
    for (int s = 0; s < std::max(len0, len1); ++s) {
        const int k0 = s < len0 ? s : -1, k1 = s < len1 ? F.K - 1 - s : -1;
        NRS_TRY(kft_invert(c, *H, k0, k1, d.flags));
        const dim3 g((F.ld + 255) / 256, F.nfm, 2), g2((F.ld + KFT_TC - 1) / KFT_TC, 2);
        const size_t shy = sizeof(double) * KFT_TC * F.ld;
        hipLaunchKernelGGL(k_kft_gct, g, dim3(256), 0, c->stream, F, k0, k1);
        if (k0 >= 0 && k1 >= 0 && k0 + 1 == k1 - 1) {              // both chains reach the middle keyframe: one after the other (fixed order)
            hipLaunchKernelGGL(k_kft_tgt, g2, dim3(KFT_TGT_NT), shy, c->stream, F, k0, -1);
            hipLaunchKernelGGL(k_kft_tgt, g2, dim3(KFT_TGT_NT), shy, c->stream, F, -1, k1);
        } else hipLaunchKernelGGL(k_kft_tgt, g2, dim3(KFT_TGT_NT), shy, c->stream, F, k0, k1);
    }
    NRS_TRY(kft_invert(c, *H, F.m, -1, d.flags));
    NRS_HIP(c, hipGetLastError());
    H->factorisations++;
    return NRS_OK;
}

// u = M^-1 r: r in (rv rows, rp poses), u to (uv rows, up poses)
// The factorisation is exact up to rounding, so the first PCG step as a rule IS the solution.  The PCG's own test -- r.u <= rtol^2 r_0.u_0 with
// u = M^-1 r -- costs a second pass over the keyframe chains (42 launches) and the first two launches of an iteration that only finds r.u
// small.  This kernel tests the step's residual on its own first, |r| <= rtol |b| over the factorisation's unknowns (fixed-order sums, one
// workgroup): when it holds the solve is marked converged -- with the iterate the PCG would have returned an iteration later -- and the
// second pass is never enqueued; when it does not, the PCG carries on as before (pcg_enqueue_batch, nrs_engine.hip).
__global__ __launch_bounds__(1024) void k_kft_rnorm(KftDev F, Dev P, const double* __restrict__ rv, const double* __restrict__ rp, double tol2, int pub_seq) {
    __shared__ double lds[2 * 16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double sr = 0, sb = 0;
    const bool live = P.flags[0] == 0;
    if (live)
        for (int k = 0; k < F.K; ++k) {
            const int nf = F.kf_nf[k], np = F.kf_np[k];
            for (int jx = tid; jx < 3 * nf + np; jx += 1024) {
                double r, b;
                if (jx < 3 * nf) {
                    const size_t o = 3 * (size_t)F.kf_row[k * F.nfm + jx / 3] + jx % 3;
                    r = rv[o]; b = P.bl[o];
                } else { r = rp[6 * k + (jx - 3 * nf)]; b = P.bp[6 * k + (jx - 3 * nf)]; }
                sr += r * r; sb += b * b;
            }
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sr += __shfl_xor(sr, o, 64); sb += __shfl_xor(sb, o, 64); }
    if (lane == 0) { lds[2 * w] = sr; lds[2 * w + 1] = sb; }
    __syncthreads();
    if (tid == 0) {
        sr = 0; sb = 0;
        for (int q = 0; q < 16; ++q) { sr += lds[2 * q]; sb += lds[2 * q + 1]; }
        if (live && sr <= tol2 * sb) {                             // (a NaN fails the test: the PCG's own checks see it)
            __threadfence();
            P.flags[0] = 1;
        }
        if (pub_seq != 0) publish_flags(P, pub_seq);
    }
}

static int kft_apply(nrs_ctx* c, KftHost* H, const double* rv, const double* rp, double* uv, double* up, const int* flags) {
    const KftDev& F = H->d;
    const int len0 = F.m, len1 = F.K - 1 - F.m, ns = std::max(len0, len1);
    const dim3 b(256);
    const size_t shm = sizeof(double) * F.ld;
    auto stage = [&](int mode, int k0, int k1) {
        const int nch = k1 >= 0 ? 2 : 1;
        hipLaunchKernelGGL(k_kft_vec, dim3((F.ld + 255) / 256, nch), b, 0, c->stream, F, mode, k0, k1, rv, rp, flags);
        hipLaunchKernelGGL(k_kft_gemv, dim3(F.ld / 16, nch), b, shm, c->stream, F, mode, k0, k1, uv, up, flags);
    };
    for (int s = 0; s < ns; ++s) stage(0, s < len0 ? s : -1, s < len1 ? F.K - 1 - s : -1);
    stage(1, F.m, -1);
    for (int s = ns - 1; s >= 0; --s) stage(2, s < len0 ? s : -1, s < len1 ? F.K - 1 - s : -1);
    return NRS_OK;
}

static int kft_setup(nrs_ctx* c, Engine* e, const EngineSpec& s, const std::vector<int>& pose_grp_ptr);   // nrs_engine_kft_setup.hpp

}  // namespace nrs
