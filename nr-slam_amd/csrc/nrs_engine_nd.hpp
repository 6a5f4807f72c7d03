// Direct solve of a2's single-frame system (H + lambda I) x = b: multifrontal Cholesky on the nested-dissection plan of
// nrs_nd_plan.hpp, dense fronts on v_mfma_f64_16x16x4.  Part of nrs_engine.hip (one translation unit).
//
// Reference: LinearSolverEigen::solve (third_party/g2o/g2o/solvers/eigen/linear_solver_eigen.h:92-136) -- a sparse Cholesky of
// the whole system per LM trial, `not positive definite` reported as a failed solve -- as CameraPoseAndDeformationOptimization
// drives it (modules/optimization/g2o_optimization.cc:148-557, block_solver.hpp:329-341: no Schur ordering, nothing marginalised).
//
// One launch per tree level (leaves first).  A workgroup is (front f, boundary row blocks I >= J): it assembles the front's
// own block F11 (<= 96 x 96) and the two 48-row blocks of F21 in LDS -- original entries, then the children's Schur
// complements through the plan's maps, in a fixed order (no atomics: bit-reproducible) -- factorises the tall panel
// [F11; F21_I; F21_J] by 16-column steps (diagonal block in one wave on cross-lane reads, panel rows one per thread, trailing
// update on the matrix cores), and leaves the tile U_IJ = F22_IJ - L21_I L21_J^T (matrix cores) for the parent.  F11 is
// factorised redundantly by every workgroup of a front: it is the latency of the level either way, and the tiles of a large
// boundary then spread over the CUs without a second launch.  The right-hand side is one more boundary row, so the forward
// substitution rides along; k_nd_back walks the levels back down (L11^T x = y - L21^T x_bnd).
#pragma once
#include "nrs_nd_plan.hpp"

namespace nrs {

constexpr int ND_LD = 97;            // LDS leading dimension (doubles): odd, so the column-strided operand reads of the MFMAs are conflict-free
constexpr int ND_S16 = 96;
typedef double nd_v4d __attribute__((ext_vector_type(4)));

struct NdDev {
    const NdFrontD* fr; const int* own; const int* bnd; const int* child; const int16_t* cmap; const NdEnt* ent; const int* wg; const int* lvl_fronts;
    const double* Dn; const double* Vp; const double* bn;          // blocks of the current linearisation (diagonal 9 / node, pairs 9, rhs 3)
    double* Lp; double* U; double* xn;
    const int* node_out;             // engine: node -> 3 doubles at out_rows + o (o >= 0) or out_pose - 1 - o (o < 0); null: xn only
    double* out_rows; double* out_pose;
    int* flags;                      // [0] done [1] iterations [2] not positive definite (the engine's PCG flags, or a scratch word block)
    long long* clk;                  // NRS_ND_DBG: 8 phase clocks (100 MHz) per workgroup of the factorisation, then per front of the back substitution; else null
};

__device__ inline double nd_readlane(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ inline double nd_rsqrt(double a) {                       // a > 0: v_rsq_f64 + two Newton steps
    double r = __builtin_amdgcn_rsq(a);
    const double h = 0.5 * a;
    r = r * (1.5 - h * r * r);
    r = r * (1.5 - h * r * r);
    return r;
}

__global__ __launch_bounds__(256) void k_nd_level(NdDev N, int wg0, double lam) {
    extern __shared__ double sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int* wd = N.wg + 3 * (size_t)(wg0 + blockIdx.x);
    const int f = wd[0], I = wd[1], J = wd[2];
    const NdFrontD F = N.fr[f];
    const int s = F.s, s16 = (s + 15) & ~15, b1 = F.b + 1, m = s + F.b, sn = s / 3, mn = m / 3;
    const int rI = min(ND_TB, b1 - ND_TB * I);                      // rows of block I (the last block is partial; J < I is always full)
    const bool two = J != I;
    const int cJ = two ? ND_TB : rI;
    const int nrow = s16 + ND_TB + (two ? ND_TB : 0);
    const int rowI0 = s16, rowJ0 = two ? s16 + ND_TB : s16;
    double* W = sm;
    double* dinv = W + (size_t)nrow * ND_LD;
    int16_t* cmo = reinterpret_cast<int16_t*>(dinv + ND_S16);
    int16_t* cmi = cmo + 32;
    int16_t* cmj = cmi + 16;
    const int tx = tid & 31, ty = tid >> 5;
    auto stamp = [&](int k) { if (N.clk && tid == 0) N.clk[8 * (size_t)(wg0 + blockIdx.x) + k] = wall_clock64(); };
    stamp(0);
    // ---- assemble: zero, original entries, children (fixed order)
    for (int i = tid; i < nrow * ND_LD; i += 256) W[i] = 0.0;
    __syncthreads();
    if (tid < s16 - s) W[(s + tid) * ND_LD + s + tid] = 1.0;       // padding columns: unit diagonal
    for (int e = tid; e < F.n_ent; e += 256) {
        const NdEnt E = N.ent[F.ent_off + e];
        const uint32_t kind = E.src >> ND_KIND_SHIFT, src = E.src & ND_SRC_MASK;
        const int fr_row = 3 * (int)E.r;
        int wr;
        if (fr_row < s) wr = fr_row;
        else {
            const int rb = fr_row - s;
            if (rb >= ND_TB * I && rb < ND_TB * I + ND_TB) wr = rowI0 + rb - ND_TB * I;
            else if (two && rb >= ND_TB * J && rb < ND_TB * J + ND_TB) wr = rowJ0 + rb - ND_TB * J;
            else continue;
        }
        double* dst = W + (size_t)wr * ND_LD + 3 * (int)E.c;
        if (kind == 2) {
            const double* v = N.bn + 3 * (size_t)src;
            dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2];
        } else {
            const double* v = (kind == 0 ? N.Dn : N.Vp) + 9 * (size_t)src;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int j = 0; j < 3; ++j) dst[a * ND_LD + j] = v[3 * a + j] + ((kind == 0 && a == j) ? lam : 0.0);
        }
    }
    stamp(1);
    nd_v4d acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[q] = nd_v4d{0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < F.n_ch; ++k) {
        const NdFrontD C = N.fr[N.child[F.ch_off + k]];
        const int16_t* cm = N.cmap + F.cmap_off + (size_t)k * (mn + 1);
        __syncthreads();
        if (tid < sn) cmo[tid] = cm[tid];
        if (tid >= 64 && tid < 80) { const int np = sn + 16 * I + (tid - 64); cmi[tid - 64] = np <= mn ? cm[np] : (int16_t)-1; }
        if (tid >= 128 && tid < 144) { const int np = sn + 16 * J + (tid - 128); cmj[tid - 128] = np <= mn ? cm[np] : (int16_t)-1; }
        __syncthreads();
        const double* Uc = N.U + C.U_off;
        const int ldc = C.ldU;
        for (int p = ty; p < s; p += 8) {                          // F11, lower triangle
            const int a = cmo[p / 3];
            if (a < 0) continue;
            const double* ur = Uc + (size_t)(3 * a + p % 3) * ldc;
            for (int q = tx; q <= p; q += 32) { const int bq = cmo[q / 3]; if (bq >= 0) W[p * ND_LD + q] += ur[3 * bq + q % 3]; }
        }
        for (int r = ty; r < rI; r += 8) {                         // F21, block I
            const int a = cmi[r / 3];
            if (a < 0) continue;
            const double* ur = Uc + (size_t)(3 * a + r % 3) * ldc;
            for (int q = tx; q < s; q += 32) { const int bq = cmo[q / 3]; if (bq >= 0) W[(rowI0 + r) * ND_LD + q] += ur[3 * bq + q % 3]; }
        }
        if (two)
            for (int r = ty; r < ND_TB; r += 8) {                  // F21, block J
                const int a = cmj[r / 3];
                if (a < 0) continue;
                const double* ur = Uc + (size_t)(3 * a + r % 3) * ldc;
                for (int q = tx; q < s; q += 32) { const int bq = cmo[q / 3]; if (bq >= 0) W[(rowJ0 + r) * ND_LD + q] += ur[3 * bq + q % 3]; }
            }
        const int16_t* cmc = two ? cmj : cmi;                      // F22 tile (I, J): straight into the accumulators
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) {
            const int t = wave + 4 * t3;
            if (t >= 9) break;
            const int ti = t / 3, tj = t % 3;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = 16 * ti + (lane >> 4) + 4 * g, cc = 16 * tj + (lane & 15);
                if (r < rI && cc < cJ && ND_TB * J + cc < F.b) {
                    const int a = cmi[r / 3], bq = cmc[cc / 3];
                    if (a >= 0 && bq >= 0) acc[t3][g] += Uc[(size_t)(3 * a + r % 3) * ldc + 3 * bq + cc % 3];
                }
            }
        }
    }
    __syncthreads();
    stamp(2);
    // ---- panel factorisation of [F11; F21_I; F21_J] by 16-column steps
    const int nb = s16 >> 4, nrt = nrow >> 4;
    int bad = 0;
    for (int kb = 0; kb < nb; ++kb) {
        const int k0 = 16 * kb;
        if (wave == 0) {                                           // diagonal block: lane i (< 16) holds row i
            const int i = lane & 15;
            double a[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) a[j] = W[(k0 + i) * ND_LD + k0 + j];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                double ajj = nd_readlane(a[j], j);
                if (!(ajj > 0.0)) { bad = 1; ajj = 1.0; }
                const double r = nd_rsqrt(ajj);
                const double lij = (i == j) ? ajj * r : a[j] * r;
                a[j] = lij;
                if (lane == 0) dinv[k0 + j] = r;
#pragma unroll
                for (int k = j + 1; k < 16; ++k) a[k] -= lij * nd_readlane(lij, k);
            }
            if (lane < 16) {
#pragma unroll
                for (int j = 0; j < 16; ++j) if (j <= i) W[(k0 + i) * ND_LD + k0 + j] = a[j];
            }
        }
        __syncthreads();
        {                                                          // panel: one row per thread, forward substitution against the block
            const int row = k0 + 16 + tid;
            if (row < nrow) {
                double x[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) x[q] = W[row * ND_LD + k0 + q];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    double v = x[q];
#pragma unroll
                    for (int p = 0; p < q; ++p) v -= x[p] * W[(k0 + q) * ND_LD + k0 + p];
                    x[q] = v * dinv[k0 + q];
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) W[row * ND_LD + k0 + q] = x[q];
            }
        }
        __syncthreads();
        {                                                          // trailing update on the matrix cores: C_rb,cb -= P_rb P_cb^T
            int cnt = 0;
            for (int cb = kb + 1; cb < nb; ++cb)
                for (int rb = cb; rb < nrt; ++rb, ++cnt) {
                    if ((cnt & 3) != wave) continue;
                    nd_v4d c;
#pragma unroll
                    for (int g = 0; g < 4; ++g) c[g] = W[(16 * rb + (lane >> 4) + 4 * g) * ND_LD + 16 * cb + (lane & 15)];
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const double av = -W[(16 * rb + (lane & 15)) * ND_LD + k0 + 4 * kk + (lane >> 4)];
                        const double bv = W[(16 * cb + (lane & 15)) * ND_LD + k0 + 4 * kk + (lane >> 4)];
                        c = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c, 0, 0, 0);
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) W[(16 * rb + (lane >> 4) + 4 * g) * ND_LD + 16 * cb + (lane & 15)] = c[g];
                }
        }
        __syncthreads();
    }
    stamp(3);
    // ---- Schur tile: U_IJ = F22_IJ - L21_I L21_J^T
#pragma unroll
    for (int t3 = 0; t3 < 3; ++t3) {
        const int t = wave + 4 * t3;
        if (t >= 9) break;
        const int ti = t / 3, tj = t % 3;
        nd_v4d c = acc[t3];
        for (int kk = 0; kk < (s16 >> 2); ++kk) {
            const double av = -W[(rowI0 + 16 * ti + (lane & 15)) * ND_LD + 4 * kk + (lane >> 4)];
            const double bv = W[(rowJ0 + 16 * tj + (lane & 15)) * ND_LD + 4 * kk + (lane >> 4)];
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c, 0, 0, 0);
        }
        double* Uf = N.U + F.U_off;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int r = 16 * ti + (lane >> 4) + 4 * g, cc = 16 * tj + (lane & 15);
            const int R = ND_TB * I + r, Cc = ND_TB * J + cc;
            if (r < rI && cc < cJ && Cc < F.b) {
                Uf[(size_t)R * F.ldU + Cc] = c[g];
                if (two && R < F.b) Uf[(size_t)Cc * F.ldU + R] = c[g];
            }
        }
    }
    stamp(4);
    // ---- the factor: block I's rows of L21 (and y^T) by the workgroups of column 0, L11 and 1 / diag by (0, 0)
    if (J == 0) {
        double* L = N.Lp + F.L_off;
        for (int r = ty; r < rI; r += 8)
            for (int q = tx; q < s; q += 32) L[(size_t)(s + ND_TB * I + r) * s + q] = W[(rowI0 + r) * ND_LD + q];
        if (I == 0) {
            for (int p = ty; p < s; p += 8)
                for (int q = tx; q < s; q += 32) L[(size_t)p * s + q] = q <= p ? W[p * ND_LD + q] : 0.0;
            if (tid < s) L[(size_t)(m + 1) * s + tid] = dinv[tid];
            if (bad && lane == 0) N.flags[2] = 1;                  // (wave 0 saw the pivots)
        }
    }
    stamp(5);
}

// back substitution of one level (the root's first): L11^T x_own = y - L21^T x_bnd, one workgroup per front
__global__ __launch_bounds__(256) void k_nd_back(NdDev N, int lf0, int fin, int clk0) {
    extern __shared__ double sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const NdFrontD F = N.fr[N.lvl_fronts[lf0 + blockIdx.x]];
    const int s = F.s, b = F.b, m = s + b;
    double* Ls = sm;                                               // L11, [s][ND_LD]
    double* part = Ls + ND_S16 * ND_LD;                            // [2][128]
    double* di = part + 256;                                       // [96]
    double* xb = di + ND_S16;                                      // [b]
    const double* L = N.Lp + F.L_off;
    auto stamp = [&](int k) { if (N.clk && tid == 0) N.clk[8 * (size_t)(clk0 + lf0 + blockIdx.x) + k] = wall_clock64(); };
    stamp(0);
    for (int i = tid; i < b; i += 256) xb[i] = N.xn[3 * (size_t)N.bnd[F.bnd_off + i / 3] + i % 3];
    for (int i = tid; i < s * s; i += 256) { const int p = i / s, q = i - p * s; Ls[p * ND_LD + q] = L[i]; }
    if (tid < s) di[tid] = L[(size_t)(m + 1) * s + tid];
    __syncthreads();
    stamp(1);
    {
        const int q = tid & 127, h = tid >> 7;
        double acc = 0;
        if (q < s)
            for (int r = h; r < b; r += 2) acc += L[(size_t)(s + r) * s + q] * xb[r];
        part[h * 128 + q] = acc;
    }
    __syncthreads();
    stamp(2);
    if (wave == 0) {
        double t0 = lane < s ? L[(size_t)m * s + lane] - part[lane] - part[128 + lane] : 0.0;
        double t1 = lane + 64 < s ? L[(size_t)m * s + lane + 64] - part[lane + 64] - part[192 + lane] : 0.0;
        double x0 = 0, x1 = 0;
        for (int p = s - 1; p >= 0; --p) {
            const double xp = (p < 64 ? nd_readlane(t0, p) : nd_readlane(t1, p - 64)) * di[p];
            if (lane == (p & 63)) { if (p < 64) x0 = xp; else x1 = xp; }
            if (lane < p) t0 -= Ls[p * ND_LD + lane] * xp;
            if (lane + 64 < p) t1 -= Ls[p * ND_LD + lane + 64] * xp;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q = lane + 64 * h;
            if (q >= s) continue;
            const double xv = h ? x1 : x0;
            const int node = N.own[F.own_off + q / 3];
            N.xn[3 * (size_t)node + q % 3] = xv;
            if (N.node_out) {
                const int o = N.node_out[node];
                if (o >= 0) N.out_rows[o + q % 3] = xv; else N.out_pose[-1 - o + q % 3] = xv;
            }
        }
    }
    stamp(3);
    if (fin && blockIdx.x == 0 && tid == 0) { N.flags[1] = 1; __threadfence(); N.flags[0] = 1; }
}

// ---- host side ------------------------------------------------------------------------------------------------------
struct NdSolver {
    NdPlan plan;
    NdDev dev;
    DevBuf own;                      // everything the kernels read: plan arrays, value blocks, L, U, x ...
    DevBuf* buf = &own;              // ... in the solver's own buffer (the tap) or in the context's (engines: reused from frame to frame)
    std::vector<size_t> lvl_shm_fac, lvl_shm_back;
    bool attr_set = false;
    double* d_Dn = nullptr; double* d_Vp = nullptr; double* d_bn = nullptr;
};

static int nd_upload(nrs_ctx* c, NdSolver& S) {
    const NdPlan& P = S.plan;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += al(bytes); return o; };
    const size_t o_fr = take(sizeof(NdFrontD) * P.fr.size()), o_own = take(4 * P.own.size()), o_bnd = take(4 * std::max<size_t>(1, P.bnd.size())),
                 o_ch = take(4 * std::max<size_t>(1, P.child.size())), o_cm = take(2 * std::max<size_t>(1, P.cmap.size())), o_ent = take(sizeof(NdEnt) * P.ent.size()),
                 o_wg = take(4 * P.wg.size()), o_lf = take(4 * P.lvl_fronts.size()), o_Dn = take(72 * (size_t)P.n_nodes), o_Vp = take(72 * std::max(1, P.n_pairs)),
                 o_bn = take(24 * (size_t)P.n_nodes), o_L = take(8 * P.L_doubles), o_U = take(8 * P.U_doubles), o_x = take(24 * (size_t)P.n_nodes),
                 o_fl = take(64);
    NRS_TRY(c->ensure(*S.buf, off));
    char* base = S.buf->as<char>();
    auto up = [&](size_t o, const void* src, size_t bytes) -> int {
        if (bytes) NRS_HIP(c, hipMemcpyAsync(base + o, src, bytes, hipMemcpyHostToDevice, c->stream));
        return NRS_OK;
    };
    NRS_TRY(up(o_fr, P.fr.data(), sizeof(NdFrontD) * P.fr.size()));
    NRS_TRY(up(o_own, P.own.data(), 4 * P.own.size()));
    NRS_TRY(up(o_bnd, P.bnd.data(), 4 * P.bnd.size()));
    NRS_TRY(up(o_ch, P.child.data(), 4 * P.child.size()));
    NRS_TRY(up(o_cm, P.cmap.data(), 2 * P.cmap.size()));
    NRS_TRY(up(o_ent, P.ent.data(), sizeof(NdEnt) * P.ent.size()));
    NRS_TRY(up(o_wg, P.wg.data(), 4 * P.wg.size()));
    NRS_TRY(up(o_lf, P.lvl_fronts.data(), 4 * P.lvl_fronts.size()));
    NdDev& D = S.dev;
    memset(&D, 0, sizeof(D));
    D.fr = reinterpret_cast<const NdFrontD*>(base + o_fr); D.own = reinterpret_cast<const int*>(base + o_own); D.bnd = reinterpret_cast<const int*>(base + o_bnd);
    D.child = reinterpret_cast<const int*>(base + o_ch); D.cmap = reinterpret_cast<const int16_t*>(base + o_cm); D.ent = reinterpret_cast<const NdEnt*>(base + o_ent);
    D.wg = reinterpret_cast<const int*>(base + o_wg); D.lvl_fronts = reinterpret_cast<const int*>(base + o_lf);
    S.d_Dn = reinterpret_cast<double*>(base + o_Dn); S.d_Vp = reinterpret_cast<double*>(base + o_Vp); S.d_bn = reinterpret_cast<double*>(base + o_bn);
    D.Dn = S.d_Dn; D.Vp = S.d_Vp; D.bn = S.d_bn;
    D.Lp = reinterpret_cast<double*>(base + o_L); D.U = reinterpret_cast<double*>(base + o_U); D.xn = reinterpret_cast<double*>(base + o_x);
    D.flags = reinterpret_cast<int*>(base + o_fl);
    NRS_HIP(c, hipMemsetAsync(base + o_fl, 0, 64, c->stream));
    // dynamic LDS per level: the largest panel / boundary of its fronts
    S.lvl_shm_fac.assign(P.n_levels, 0); S.lvl_shm_back.assign(P.n_levels, 0);
    for (int l = 0; l < P.n_levels; ++l)
        for (int i = P.lvl_ptr[l]; i < P.lvl_ptr[l + 1]; ++i) {
            const NdFrontD& F = P.fr[P.lvl_fronts[i]];
            const int s16 = (F.s + 15) & ~15, nrow = s16 + ND_TB + (F.nR > 1 ? ND_TB : 0);
            S.lvl_shm_fac[l] = std::max(S.lvl_shm_fac[l], sizeof(double) * ((size_t)nrow * ND_LD + ND_S16) + 2 * 64);
            S.lvl_shm_back[l] = std::max(S.lvl_shm_back[l], sizeof(double) * ((size_t)ND_S16 * ND_LD + 256 + ND_S16 + (size_t)F.b));
        }
    if (!S.attr_set) {
        NRS_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_nd_level), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        NRS_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_nd_back), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        S.attr_set = true;
    }
    for (int l = 0; l < P.n_levels; ++l)
        if (S.lvl_shm_back[l] > 160 * 1024) return c->fail(NRS_ERR_INVALID, "direct solve: a front's boundary does not fit the back substitution's LDS");
    return NRS_OK;
}

// factorise (H + lam I) and solve: 2 x levels launches on the context's stream, no host synchronisation
static int nd_solve_enqueue(nrs_ctx* c, NdSolver& S, double lam) {
    const NdPlan& P = S.plan;
    for (int l = 0; l < P.n_levels; ++l) {
        const int n = P.lvl_wg_ptr[l + 1] - P.lvl_wg_ptr[l];
        hipLaunchKernelGGL(k_nd_level, dim3(n), dim3(256), S.lvl_shm_fac[l], c->stream, S.dev, P.lvl_wg_ptr[l], lam);
    }
    for (int l = P.n_levels - 1; l >= 0; --l)
        hipLaunchKernelGGL(k_nd_back, dim3(P.lvl_ptr[l + 1] - P.lvl_ptr[l]), dim3(256), S.lvl_shm_back[l], c->stream, S.dev, P.lvl_ptr[l], l == 0 ? 1 : 0, (int)P.wg.size() / 3);
    NRS_HIP(c, hipGetLastError());
    return NRS_OK;
}

void nd_orient_pairs(const NdPlan& P, const int32_t* pairs, const double* Vp, std::vector<double>& out);   // nrs_host_build.cpp
void nd_stats(const NdPlan& P, int64_t* stats);

// include/nrs.h nrs_debug_nd_solve
int engine_nd_debug_solve(nrs_ctx* c, int n_nodes, const double* pos, const uint8_t* last, int n_pairs, const int* pairs, const double* Dn, const double* Vp,
                          const double* bn, double lam, int repeats, double* x, int64_t* stats, double* ms_per_solve) {
    NRS_HIP(c, hipSetDevice(c->device));
    NdSolver S;
    std::string err;
    if (!nd_build_plan(n_nodes, pos, last, n_pairs, pairs, S.plan, &err)) return c->fail(NRS_ERR_INVALID, "direct solve: %s", err.c_str());
    nd_stats(S.plan, stats);
    struct Rel { nrs_ctx* c; NdSolver* s; ~Rel() { (void)hipStreamSynchronize(c->stream); c->release(s->own); } } rel{c, &S};
    NRS_TRY(nd_upload(c, S));
    std::vector<double> V;
    nd_orient_pairs(S.plan, pairs, Vp, V);
    NRS_HIP(c, hipMemcpyAsync(S.d_Dn, Dn, 72 * (size_t)n_nodes, hipMemcpyHostToDevice, c->stream));
    if (n_pairs) NRS_HIP(c, hipMemcpyAsync(S.d_Vp, V.data(), 72 * (size_t)n_pairs, hipMemcpyHostToDevice, c->stream));
    NRS_HIP(c, hipMemcpyAsync(S.d_bn, bn, 24 * (size_t)n_nodes, hipMemcpyHostToDevice, c->stream));
    NRS_TRY(nd_solve_enqueue(c, S, lam));                          // (warm-up and the result)
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    if (repeats > 0) {
        NRS_HIP(c, hipEventRecord(c->ev0, c->stream));
        for (int r = 0; r < repeats; ++r) NRS_TRY(nd_solve_enqueue(c, S, lam));
        NRS_HIP(c, hipEventRecord(c->ev1, c->stream));
        NRS_HIP(c, hipEventSynchronize(c->ev1));
        float ms = 0;
        NRS_HIP(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
        if (ms_per_solve) *ms_per_solve = ms / repeats;
    }
    if (getenv("NRS_ND_DBG")) {                                    // phase clocks of one solve: mean / max over the workgroups of every launch
        const size_t nw = S.plan.wg.size() / 3 + (size_t)S.plan.n_fronts;
        long long* clk = nullptr;
        NRS_HIP(c, hipMalloc((void**)&clk, sizeof(long long) * 8 * nw));
        NRS_HIP(c, hipMemsetAsync(clk, 0, sizeof(long long) * 8 * nw, c->stream));
        S.dev.clk = clk;
        NRS_TRY(nd_solve_enqueue(c, S, lam));
        S.dev.clk = nullptr;
        std::vector<long long> h(8 * nw);
        NRS_HIP(c, hipMemcpyAsync(h.data(), clk, sizeof(long long) * 8 * nw, hipMemcpyDeviceToHost, c->stream));
        NRS_HIP(c, hipStreamSynchronize(c->stream));
        (void)hipFree(clk);
        const NdPlan& P = S.plan;
        const long long t00 = h[0];
        for (int l = 0; l < P.n_levels; ++l) {
            double mean[5] = {0, 0, 0, 0, 0}, mx[5] = {0, 0, 0, 0, 0};
            long long lo = LLONG_MAX, hi = 0;
            const int a = P.lvl_wg_ptr[l], b2 = P.lvl_wg_ptr[l + 1];
            for (int w = a; w < b2; ++w) {
                const long long* q = &h[8 * (size_t)w];
                for (int k = 0; k < 5; ++k) { const double d = (double)(q[k + 1] - q[k]) / 100.0; mean[k] += d / (b2 - a); mx[k] = std::max(mx[k], d); }
                lo = std::min(lo, q[0]); hi = std::max(hi, q[5]);
            }
            fprintf(stderr, "[nrs] nd level %2d: %4d wg, span %6.1f us (from %7.1f) | mean / max us: entries %.1f/%.1f gather %.1f/%.1f factor %.1f/%.1f schur %.1f/%.1f store %.1f/%.1f\n", l, b2 - a,
                    (double)(hi - lo) / 100.0, (double)(lo - t00) / 100.0, mean[0], mx[0], mean[1], mx[1], mean[2], mx[2], mean[3], mx[3], mean[4], mx[4]);
        }
        for (int l = P.n_levels - 1; l >= 0; --l) {
            double mean[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
            long long lo = LLONG_MAX, hi = 0;
            const int a = P.lvl_ptr[l], b2 = P.lvl_ptr[l + 1];
            for (int w = a; w < b2; ++w) {
                const long long* q = &h[8 * (P.wg.size() / 3 + (size_t)w)];
                for (int k = 0; k < 3; ++k) { const double d = (double)(q[k + 1] - q[k]) / 100.0; mean[k] += d / (b2 - a); mx[k] = std::max(mx[k], d); }
                lo = std::min(lo, q[0]); hi = std::max(hi, q[3]);
            }
            fprintf(stderr, "[nrs] nd back  %2d: %4d wg, span %6.1f us (from %7.1f) | mean / max us: loads %.1f/%.1f gemv %.1f/%.1f solve %.1f/%.1f\n", l, b2 - a,
                    (double)(hi - lo) / 100.0, (double)(lo - t00) / 100.0, mean[0], mx[0], mean[1], mx[1], mean[2], mx[2]);
        }
    }
    int fl[4] = {0, 0, 0, 0};
    NRS_HIP(c, hipMemcpyAsync(fl, S.dev.flags, sizeof(fl), hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipMemcpyAsync(x, S.dev.xn, 24 * (size_t)n_nodes, hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    if (!fl[0]) return c->fail(NRS_ERR_HIP, "direct solve: the last level did not report completion");
    return fl[2] ? c->fail(NRS_ERR_NUMERIC, "direct solve: the matrix is not positive definite") : NRS_OK;
}

// ---- engine glue: a2's single-frame engines (K = 1) on the direct path -----------------------------------------------
// Nodes = the free rows (3 unknowns each) + the two halves of the pose block when the pose is free.  Per linearisation
// k_nd_values turns what the lineariser left (row diagonal blocks D, gradients, H_pp / b_p, the per-incidence factors of the
// springs and dampers, the 32-byte reprojection factors of the rows) into explicit blocks: the pair block of two coupled rows
// is -(sum qc v v^T + sum s I) over the edges that join them (v = x_i - x_j at the linearisation point, exactly what the
// factored operator of the PCG path applies), a pose-row block is J_p^T w J_l rebuilt from the row's fp32 projection Jacobian.
struct NdPairD { int kind, a, b, src0, nsrc; };        // kind 0: rows (a, b), sources src[src0 .. src0 + nsrc); 1: (pose half a, row b); 2: the pose's off-diagonal block
struct NdVals {
    const int* node_row;             // node -> row (>= 0) or -1 - half
    const NdPairD* pair;
    const int* src;                  // (incidence slot << 1) | (0 spring, 1 damper)
    double* Dn; double* Vp; double* bn;
    int n_nodes, n_pairs;
};

__global__ __launch_bounds__(256) void k_nd_values(Dev P, NdVals V) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < V.n_nodes) {
        const int row = V.node_row[i];
        double* d = V.Dn + 9 * (size_t)i;
        double* g = V.bn + 3 * (size_t)i;
        if (row >= 0) {
            const double* D = P.D + 6 * (size_t)row;
            d[0] = D[0]; d[1] = D[1]; d[2] = D[2]; d[3] = D[1]; d[4] = D[3]; d[5] = D[4]; d[6] = D[2]; d[7] = D[4]; d[8] = D[5];
            g[0] = P.bl[3 * (size_t)row]; g[1] = P.bl[3 * (size_t)row + 1]; g[2] = P.bl[3 * (size_t)row + 2];
        } else {
            const int h = -1 - row;                                // H_pp is packed upper-triangular, 21 entries (pose 0)
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const int r = 3 * h + min(a, b), cc = 3 * h + max(a, b);
                    d[3 * a + b] = P.Hpp[r * 6 - (r * (r - 1)) / 2 + (cc - r)];
                }
            g[0] = P.bp[3 * h]; g[1] = P.bp[3 * h + 1]; g[2] = P.bp[3 * h + 2];
        }
    }
    if (i < V.n_pairs) {
        const NdPairD q = V.pair[i];
        double* o = V.Vp + 9 * (size_t)i;
        if (q.kind == 0) {
            double v[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                v[k] = P.lin_xl[3 * (size_t)q.a + k] - P.lin_xl[3 * (size_t)q.b + k];
                if (P.X0) v[k] = (P.lin_xl[3 * (size_t)q.a + k] + P.X0[3 * (size_t)q.a + k]) - (P.lin_xl[3 * (size_t)q.b + k] + P.X0[3 * (size_t)q.b + k]);
            }
            double qc = 0, sd = 0;
            for (int k = 0; k < q.nsrc; ++k) {
                const int sv = V.src[q.src0 + k];
                if (sv & 1) sd += P.d_s[sv >> 1]; else qc += P.s_qc[sv >> 1];
            }
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) o[3 * a + b] = -(qc * v[a] * v[b] + (a == b ? sd : 0.0));
        } else if (q.kind == 1) {
            // H_{pose half, row} = J_p^T w J_l, J_l = -J R, J_p = -J [-[X_c]x | I] (reprojection_error_with_deformation.cc:52-68), as row_factored() forms them
            const RowRec rc = P.rowrec[q.b];
            const Pose Tcw = P.lin_pose[0];
            double R[9];
            quat_to_R(Tcw.q, R);
            double xs[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) xs[k] = P.lin_xl[3 * (size_t)q.b + k] + (P.X0 ? P.X0[3 * (size_t)q.b + k] : 0.0);
            const double px = R[0] * xs[0] + R[1] * xs[1] + R[2] * xs[2] + Tcw.t[0];
            const double py = R[3] * xs[0] + R[4] * xs[1] + R[5] * xs[2] + Tcw.t[1];
            const double pz = R[6] * xs[0] + R[7] * xs[1] + R[8] * xs[2] + Tcw.t[2];
            double Jl[2][3], Jp[2][3];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const double j0 = -(double)rc.J[3 * rr], j1 = -(double)rc.J[3 * rr + 1], j2 = -(double)rc.J[3 * rr + 2];
                if (q.a == 0) { Jp[rr][0] = -j1 * pz + j2 * py; Jp[rr][1] = j0 * pz - j2 * px; Jp[rr][2] = -j0 * py + j1 * px; }
                else { Jp[rr][0] = j0; Jp[rr][1] = j1; Jp[rr][2] = j2; }
                Jl[rr][0] = j0 * R[0] + j1 * R[3] + j2 * R[6];
                Jl[rr][1] = j0 * R[1] + j1 * R[4] + j2 * R[7];
                Jl[rr][2] = j0 * R[2] + j1 * R[5] + j2 * R[8];
            }
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) o[3 * a + b] = rc.w * (Jp[0][a] * Jl[0][b] + Jp[1][a] * Jl[1][b]);
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) { const int r = b, cc = 3 + a; o[3 * a + b] = P.Hpp[r * 6 - (r * (r - 1)) / 2 + (cc - r)]; }   // rows: the second half
        }
    }
}

struct NdEngine {
    NdSolver S;
    NdVals vals;
    std::vector<double> pos;         // row positions the dissection was built on (rebuilt when the fixed set changes)
    std::vector<uint8_t> sig;        // RF_FIXED of every vertex + the pose's flag at set-up
    bool on = false;
};

static bool nd_wanted(nrs_ctx* c, const Dev& d, int n_free) {
    // nrs_options.direct_solve (0: by size, 1: whenever possible, 2: never); NRS_ND / NRS_ND_MAX_ROWS override it for experiments
    int mode = c->opt.direct_solve;
    if (const char* ev = getenv("NRS_ND")) mode = atoi(ev) ? 1 : 2;
    const int nmax = getenv("NRS_ND_MAX_ROWS") ? atoi(getenv("NRS_ND_MAX_ROWS")) : 2600;
    if (mode == 2 || d.K != 1 || !d.use_lds || d.dform || d.sh_on) return false;
    if (mode == 1) return n_free > 0;
    return n_free >= 48 && n_free <= nmax;
}

// builds (or rebuilds) the plan for the engine's current fixed set; leaves nd->on = false if the problem does not qualify
static int nd_engine_setup(nrs_ctx* c, Engine* e, NdEngine* nd) {
    Dev& d = e->d;
    nd->on = false;
    std::vector<int> node_of(d.M, -1), node_row;
    for (int v = 0; v < d.M; ++v)
        if (!(e->h_rflag[e->vrow[v]] & RF_FIXED)) { node_of[v] = (int)node_row.size(); node_row.push_back(e->vrow[v]); }
    const int n_free = (int)node_row.size();
    if (!nd_wanted(c, d, n_free)) return NRS_OK;
    for (size_t q = 0; q < e->dm_idx.size(); q += 4)
        if (e->dm_idx[q] >= 0 || e->dm_idx[q + 1] >= 0) return NRS_OK;           // four-vertex dampers: a BA window, not this solver's problem
    const bool pose_free = !e->h_pose_fixed[0];
    const int n_nodes = n_free + (pose_free ? 2 : 0);
    // unique row-row couplings with the incidence slots that contribute to them
    struct Key { uint64_t k; int src; };
    std::vector<Key> keys;
    keys.reserve(e->sp_ij.size() / 2 + e->dm_idx.size() / 4);
    auto add = [&](int va, int vb, int slot, int kind) {
        const int a = node_of[va], b = node_of[vb];
        if (a < 0 || b < 0 || a == b || slot < 0) return;
        keys.push_back(Key{((uint64_t)std::min(a, b) << 32) | (uint32_t)std::max(a, b), (slot << 1) | kind});
    };
    // (the factor of an edge sits in both endpoints' incidence slots with the same value when both are free: the first one is read)
    for (size_t q = 0; q < e->sp_ij.size() / 2; ++q) add(e->sp_ij[2 * q], e->sp_ij[2 * q + 1], e->sp_pos[2 * q], 0);
    for (size_t q = 0; q < e->dm_idx.size() / 4; ++q) add(e->dm_idx[4 * q + 2], e->dm_idx[4 * q + 3], e->dm_pos[4 * q + 2], 1);
    std::stable_sort(keys.begin(), keys.end(), [](const Key& x, const Key& y) { return x.k < y.k; });
    std::vector<int> pairs, src;
    std::vector<NdPairD> pd;
    for (size_t i = 0; i < keys.size();) {
        size_t j = i;
        const int a = (int)(keys[i].k >> 32), b = (int)(keys[i].k & 0xFFFFFFFFu);
        NdPairD p{0, node_row[a], node_row[b], (int)src.size(), 0};
        for (; j < keys.size() && keys[j].k == keys[i].k; ++j) src.push_back(keys[j].src);
        p.nsrc = (int)(j - i);
        pd.push_back(p);
        pairs.push_back(a); pairs.push_back(b);
        i = j;
    }
    std::vector<uint8_t> last(n_nodes, 0);
    if (pose_free) {
        last[n_free] = last[n_free + 1] = 1;
        for (int a = 0; a < n_free; ++a)
            if (e->h_rflag[node_row[a]] & RF_OBS)
                for (int h = 0; h < 2; ++h) { pd.push_back(NdPairD{1, h, node_row[a], 0, 0}); pairs.push_back(n_free + h); pairs.push_back(a); }
        pd.push_back(NdPairD{2, 0, 0, 0, 0}); pairs.push_back(n_free + 1); pairs.push_back(n_free);
    }
    std::vector<double> pos(3 * (size_t)n_nodes, 0.0);
    for (int a = 0; a < n_free; ++a)
        for (int k = 0; k < 3; ++k) pos[3 * (size_t)a + k] = nd->pos[3 * (size_t)node_row[a] + k];
    std::string err;
    const int n_pairs = (int)pd.size();
    nd->S.buf = &c->nd_ws;
    if (!nd_build_plan(n_nodes, pos.data(), last.data(), n_pairs, pairs.data(), nd->S.plan, &err)) {
        if (getenv("NRS_TIMING")) fprintf(stderr, "[nrs] direct solve not used: %s\n", err.c_str());
        return NRS_OK;
    }
    NRS_TRY(nd_upload(c, nd->S));
    std::vector<int> nrow(node_row);
    std::vector<int> node_out(n_nodes);
    for (int a = 0; a < n_free; ++a) node_out[a] = 3 * node_row[a];
    if (pose_free) { nrow.push_back(-1); nrow.push_back(-2); node_out[n_free] = -1; node_out[n_free + 1] = -1 - 3; }
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_nr = 0, o_no = o_nr + al(4 * (size_t)n_nodes), o_pd = o_no + al(4 * (size_t)n_nodes), o_src = o_pd + al(sizeof(NdPairD) * (size_t)n_pairs),
                 total = o_src + al(4 * std::max<size_t>(1, src.size()));
    NRS_TRY(c->ensure(c->nd_vals, total));
    char* vb = c->nd_vals.as<char>();
    NRS_HIP(c, hipMemcpyAsync(vb + o_nr, nrow.data(), 4 * (size_t)n_nodes, hipMemcpyHostToDevice, c->stream));
    NRS_HIP(c, hipMemcpyAsync(vb + o_no, node_out.data(), 4 * (size_t)n_nodes, hipMemcpyHostToDevice, c->stream));
    NRS_HIP(c, hipMemcpyAsync(vb + o_pd, pd.data(), sizeof(NdPairD) * (size_t)n_pairs, hipMemcpyHostToDevice, c->stream));
    if (!src.empty()) NRS_HIP(c, hipMemcpyAsync(vb + o_src, src.data(), 4 * src.size(), hipMemcpyHostToDevice, c->stream));
    nd->vals.node_row = reinterpret_cast<const int*>(vb + o_nr);
    nd->vals.pair = reinterpret_cast<const NdPairD*>(vb + o_pd);
    nd->vals.src = reinterpret_cast<const int*>(vb + o_src);
    nd->vals.Dn = nd->S.d_Dn; nd->vals.Vp = nd->S.d_Vp; nd->vals.bn = nd->S.d_bn;
    nd->vals.n_nodes = n_nodes; nd->vals.n_pairs = n_pairs;
    nd->S.dev.node_out = reinterpret_cast<const int*>(vb + o_no);
    nd->S.dev.out_rows = d.xv; nd->S.dev.out_pose = d.xp;
    nd->S.dev.flags = d.flags;
    // rows the solver never writes (fixed, padding) keep a zero step; so does a fixed pose
    NRS_HIP(c, hipMemsetAsync(d.xv, 0, sizeof(double) * 3 * (size_t)d.n_rows, c->stream));
    NRS_HIP(c, hipMemsetAsync(d.xp, 0, sizeof(double) * 6 * (size_t)d.K, c->stream));
    NRS_HIP(c, hipStreamSynchronize(c->stream));                   // the staging vectors die here
    nd->sig.assign((size_t)d.M + 1, 0);
    for (int v = 0; v < d.M; ++v) nd->sig[v] = e->h_rflag[e->vrow[v]] & RF_FIXED;
    nd->sig[d.M] = e->h_pose_fixed[0];
    nd->on = true;
    if (getenv("NRS_TIMING"))
        fprintf(stderr, "[nrs] direct solve: %d free rows, %d pairs, %d fronts on %d levels, %d workgroups, %.1f MFLOP per factorisation\n", n_free, n_pairs,
                nd->S.plan.n_fronts, nd->S.plan.n_levels, (int)nd->S.plan.wg.size() / 3, nd->S.plan.flops / 1e6);
    return NRS_OK;
}

static void nd_engine_free(NdEngine* nd) { delete nd; }          // (its device arrays live in the context's buffers)

}  // namespace nrs
