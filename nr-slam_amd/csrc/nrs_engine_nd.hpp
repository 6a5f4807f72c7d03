// Direct solve of a2's single-frame system (H + lambda I) x = b: multifrontal Cholesky on the nested-dissection plan of
// nrs_nd_plan.hpp, dense fronts on v_mfma_f64_16x16x4.  Part of nrs_engine.hip (one translation unit).
//
// Reference: LinearSolverEigen::solve (third_party/g2o/g2o/solvers/eigen/linear_solver_eigen.h:92-136) -- a sparse Cholesky of
// the whole system per LM trial, `not positive definite` reported as a failed solve -- as CameraPoseAndDeformationOptimization
// drives it (modules/optimization/g2o_optimization.cc:148-557, block_solver.hpp:329-341: no Schur ordering, nothing marginalised).
//
// Factorisation: one launch per tree level (leaves first).  A workgroup is (front f, boundary row blocks I >= J): it assembles
// the front's own block F11 (<= 96 x 96) and the two 48-row blocks of F21 in LDS -- original entries, then the children's Schur
// complements by dense reads of the slots they wrote in THIS front's index space, in a fixed order (no atomics:
// bit-reproducible) -- factorises the tall panel [F11; F21_I; F21_J] by 16-column steps (diagonal block in one wave and panel
// rows one per thread, both on DPP row broadcasts; trailing update on the matrix cores), and leaves the tile
// U_IJ = F22_IJ - L21_I L21_J^T (matrix cores) in the parent's slot.  F11 is factorised redundantly by every workgroup of a
// front: it is the latency of the level either way, and the tiles of a large boundary then spread over the CUs without a second
// launch.  The right-hand side is one more boundary row, so the forward substitution rides along.  One more workgroup per front
// factorises [F11; I] and leaves (L11^-1)^T behind the factor.
// Back pass: ONE launch (k_nd_back), a workgroup per front, top-down: factors staged on chip, then ancestor by ancestor
// (the unknowns above are polled where they land: xn is poisoned at the start of a solve) x_own = (L11^-1)^T (y - L21^T x_bnd) as
// two matrix-vector products.
// Set-up: nd_prep_run (structure only: pair lists, cache key, plan; on a helper thread of engine_create) and nd_engine_finish
// (value descriptors in the engine's row layout, uploads); the context caches the last plans (NdCache).
#pragma once
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include "nrs_nd_plan.hpp"

namespace nrs {

constexpr int ND_LD = 97;            // LDS leading dimension (doubles): odd, so the column-strided operand reads of the MFMAs are conflict-free
constexpr int ND_S16 = 96;
typedef double nd_v4d __attribute__((ext_vector_type(4)));
constexpr unsigned long long ND_POISON = 0x7FF8A5A5DEADBEEFull;   // "not written yet" in xn: a NaN payload no arithmetic produces

struct NdWgD { NdFrontD F; int I, J, pad; };          // one workgroup of k_nd_level: its front and its (I >= J) pair of row blocks
struct NdDev {
    const int* own; const int* bnd; const int* seg; const int16_t* pmap; const NdEnt* ent; const NdWgD* wg; const NdFrontD* lvl_fr;
    const double* ev;                // 9 doubles per original entry (plan order): the blocks of the current linearisation
    double* A;                       // assembly areas: every front's Schur complement lands in its parent's index space
    double* Lp; double* xn;
    const int* node_out;             // engine: node -> 3 doubles at out_rows + o (o >= 0) or out_pose - 1 - o (o < 0); null: xn only
    double* out_rows; double* out_pose;
    int* done;                       // per front: the solve (epoch) whose back substitution has written its unknowns (single-launch back pass)
    int* fcnt;                       // per front: Schur tiles its children have delivered, over all solves (single-launch factorisation)
    int* flags;                      // [0] done [1] iterations [2] not positive definite (the engine's PCG flags, or a scratch word block)
    int n_x3;                        // 3 x nodes: the length of xn
    int x_poll;                      // back pass: 1 = a front reads its boundary's unknowns by polling the VALUES (xn is poisoned when a solve starts and every
                                     // unknown is written once), 0 = by its ancestors' done flags (NRS_ND_BACK_FLAGS=1: the round-4 hand-over)
    const int* abort; int abort_id;  // speculative trials (engine_optimize): a solve whose id the host has written to *abort is not needed any more -- the
                                     // remaining workgroups of its FACTORISATION return at once (null: never); they wait for nobody, so a stale read only costs time
    long long* clk;                  // NRS_ND_DBG: 8 phase clocks (100 MHz) per workgroup of the factorisation, then per front of the back substitution; else null
};

// value of lane K of the caller's 16-lane row, in every lane of the row: DPP row_newbcast (a plain VALU move, no SGPR round trip)
template <int K>
__device__ inline double nd_rowbcast(double v) { return __builtin_amdgcn_update_dpp(v, v, 0x150 + K, 0xf, 0xf, false); }
// a += (lane K's nl of this 16-lane row) * l in ONE instruction: the DPP form of v_fmac_f64 (gfx90a+ encode row_newbcast on the
// fp64 ALU).  Measured on gfx950 (tools/micro/diag_probe.hip), per 16 x 16 block: v_readlane + FMA 5100 cycles, v_mov_b64_dpp +
// FMA 4480, this form 3350.
template <int K>
__device__ inline void nd_fmac_bcast(double& a, double nl, double l) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(nl), "v"(l), "n"(K));
}

// a -= (lane K's ls of this 16-lane row) * l: the negation rides on the DPP operand (src0 neg modifier), so no negated copy is made
template <int K>
__device__ inline void nd_fmacn_bcast(double& a, double ls, double l) {
    asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(ls), "v"(l), "n"(K));
}

// Cholesky of one 16 x 16 diagonal block of the panel, one wave: lane (i = lane & 15) of every 16-lane row holds row i (the four
// rows of the wave work redundantly, so every broadcast stays inside a row).  Column j: the pivot reaches the lanes by a row
// broadcast, a[k] -= l_ij l_kj by the DPP FMA.  Leaves the block in W (lower triangle) and 1 / diag in dinv.  The wave is ISSUE-bound
// here (tools/micro/diag_probe.hip: pinning the next column's pivot chain between this column's independent updates buys 8 %, a
// shorter chain 13 %): ten VALU operations per pivot -- a bad pivot is replaced by changing its high word only (any value in
// [1, 2) will do: one select instead of two), no negated copy of the column -- 3350 -> 2900 cycles per block.
template <int J, int K>
__device__ inline void nd_diag_cols_upd(double (&a)[16], double l) {
    if constexpr (K < 16) {
        nd_fmacn_bcast<K>(a[K], l, l);
        nd_diag_cols_upd<J, K + 1>(a, l);
    }
}
template <int J>
__device__ inline void nd_diag_cols(double (&a)[16], double (&rr)[16], int& bad) {
    if constexpr (J < 16) {
        double ajj = nd_rowbcast<J>(a[J]);
        const bool ok = ajj > 0.0;
        bad |= !ok;
        ajj = __hiloint2double(ok ? __double2hiint(ajj) : 0x3FF00000, __double2loint(ajj));
        const double r = fast_rsqrt_pos(ajj);
        double l = a[J] * r;                                       // (lane J: a_jj r = sqrt(a_jj))
        asm volatile("s_nop 1" : "+v"(l));                         // (a VALU result read through DPP needs two wait states)
        a[J] = l; rr[J] = r;
        nd_diag_cols_upd<J, J + 1>(a, l);
        nd_diag_cols<J + 1>(a, rr, bad);
    }
}

__device__ __forceinline__ void nd_diag_factor(double* W, double* dinv, int k0, int lane, int& bad) {
    const int i = lane & 15;
    double a[16], rr[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = W[(k0 + i) * ND_LD + k0 + j];
    nd_diag_cols<0>(a, rr, bad);
    if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j <= i) W[(k0 + i) * ND_LD + k0 + j] = a[j];
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) dinv[k0 + j] = rr[j];
    }
}

// step B of the panel factorisation for one row held in x: x <- x L_kk^-T, L_kk row (lane & 15) in lk (see k_nd_level)
template <int P, int Q>
__device__ inline void nd_b_upd(double (&x)[16], const double (&lk)[16], double xp) {
    if constexpr (Q < 16) {
        nd_fmacn_bcast<Q>(x[Q], lk[P], xp);                        // x[Q] -= (lane Q's L[Q][P]) * x[P]
        nd_b_upd<P, Q + 1>(x, lk, xp);
    }
}
template <int P>
__device__ inline void nd_b_cols(double (&x)[16], const double (&lk)[16], const double (&di)[16]) {
    if constexpr (P < 16) {
        x[P] *= di[P];
        nd_b_upd<P, P + 1>(x, lk, x[P]);
        nd_b_cols<P + 1>(x, lk, di);
    }
}

// the solved unknowns of a front go to the node vector and, for an engine, straight into its step vectors; the two index loads
// (node of the unknown, its output slot) are requested at kernel start (NdOut) so that no memory round trip follows the solve
struct NdOut { int node[2], o[2]; };
__device__ __forceinline__ NdOut nd_out_request(const NdDev& N, const NdFrontD& F, int lane) {
    NdOut r;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int q = lane + 64 * h;
        r.node[h] = N.own[F.own_off + (q < F.s ? q / 3 : 0)];
        r.o[h] = N.node_out ? N.node_out[r.node[h]] : 0;
    }
    return r;
}
__device__ __forceinline__ void nd_store_x(const NdDev& N, const NdFrontD& F, const NdOut& r, int lane, double x0, double x1) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int q = lane + 64 * h;
        if (q >= F.s) continue;
        const double xv = h ? x1 : x0;
        if (N.x_poll) __hip_atomic_store(N.xn + 3 * (size_t)r.node[h] + q % 3, xv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (a reader polls this very word)
        else N.xn[3 * (size_t)r.node[h] + q % 3] = xv;
        if (N.node_out) { if (r.o[h] >= 0) N.out_rows[r.o[h] + q % 3] = xv; else N.out_pose[-1 - r.o[h] + q % 3] = xv; }
    }
}

// trailing update of the panel factorisation on the matrix cores: C_rb,cb -= P_rb P_cb^T for the block columns cb in [cb_lo, cb_hi) and
// the row blocks rb >= cb, P = the 16 columns at k0; the tiles are dealt round-robin to the waves w0 .. w0 + nw - 1 (this wave: widx)
// (KB: the panel is KB consecutive 16-column blocks at k0 -- two of them in the 32-column steps, applied in column order: the same
// sequence of matrix-core operations on a tile as two single-block updates one after the other)
template <int KB = 1>
__device__ __forceinline__ void nd_update(double* W, int lane, int k0, int cb_lo, int cb_hi, int nrt, int widx, int nw, int skip_first = 0) {
    if (widx < 0 || widx >= nw) return;
    int cnt = 0;
#pragma unroll 1
    for (int cb = cb_lo; cb < cb_hi; ++cb)
#pragma unroll 1
        for (int rb = cb + (cb == cb_lo ? skip_first : 0); rb < nrt; ++rb, ++cnt) {      // (skip_first: the diagonal tile of the first column is somebody else's)
            if (cnt % nw != widx) continue;
            nd_v4d c;
            double av[4 * KB], bv[4 * KB];
#pragma unroll
            for (int g = 0; g < 4; ++g) c[g] = W[(16 * rb + (lane >> 4) + 4 * g) * ND_LD + 16 * cb + (lane & 15)];
#pragma unroll
            for (int kk = 0; kk < 4 * KB; ++kk) {
                av[kk] = -W[(16 * rb + (lane & 15)) * ND_LD + k0 + 4 * kk + (lane >> 4)];
                bv[kk] = W[(16 * cb + (lane & 15)) * ND_LD + k0 + 4 * kk + (lane >> 4)];
            }
#pragma unroll
            for (int kk = 0; kk < 4 * KB; ++kk) c = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bv[kk], c, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) W[(16 * rb + (lane >> 4) + 4 * g) * ND_LD + 16 * cb + (lane & 15)] = c[g];
        }
}

// step B of the panel factorisation for the panel row `row` against the diagonal block at k0 (one row per calling thread; the store is
// predicated by `live`): see k_nd_level
__device__ __forceinline__ void nd_b_row(double* W, const double* dinv, int k0, int row, bool live, int lane) {
    const int li = lane & 15;
    double x[16], lk[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) { x[q] = W[row * ND_LD + k0 + q]; lk[q] = W[(k0 + li) * ND_LD + k0 + q]; }
    double di[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) di[q] = dinv[k0 + q];
    nd_b_cols<0>(x, lk, di);
    if (live) {
#pragma unroll
        for (int q = 0; q < 16; ++q) W[row * ND_LD + k0 + q] = x[q];
    }
}

// NTH threads per workgroup: 256, or 512 (two waves per SIMD: seven waves instead of three on the trailing updates next to the diagonal
// block, one pass over the children's slots instead of two, a Schur tile per wave); which wave computes a tile does not change its bits
template <int NTH, bool W32 = false>
__global__ __launch_bounds__(NTH) void k_nd_level(NdDev N, int wg0, double lam, int epoch, int chained, int first) {   // first: the first launch of a solve (poisons xn for the back pass)   // chained: 0 = one launch per level, else the count of single-launch factorisations so far
    extern __shared__ double sm[];
    constexpr int NW = NTH / 64, NT3 = (9 + NW - 1) / NW;          // waves; Schur tiles (of nine) per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const NdWgD wd = N.wg[wg0 + blockIdx.x];                        // (front descriptor inlined: one scalar round trip)
    const int I = wd.I, J = wd.J;
    const NdFrontD& F = wd.F;
    const int s = F.s, s16 = (s + 15) & ~15, b1 = F.b + 1, m = s + F.b;
    // I < 0: the front's INVERSE workgroup.  Its panel is [F11; identity]: the factorisation leaves e_j^T L11^-T = row j of
    // (L11^-1)^T under F11, which the back pass multiplies with instead of substituting (no dependent chain on its critical path)
    const bool inv = I < 0;
    const int rI = inv ? 0 : min(ND_TB, b1 - ND_TB * I);           // rows of block I (the last block is partial; J < I is always full)
    const bool two = !inv && J != I;
    const int cJ = two ? ND_TB : rI;
    const int nrow = inv ? 2 * s16 : s16 + ND_TB + (two ? ND_TB : 0);
    const int rowI0 = s16, rowJ0 = two ? s16 + ND_TB : s16;
    double* W = sm;
    double* dinv = W + (size_t)nrow * ND_LD;
    int16_t* pmi = reinterpret_cast<int16_t*>(dinv + ND_S16 + 256);   // (256 doubles unused)           // parent node positions of the nodes of blocks I and J
    int16_t* pmj = pmi + 16;
    auto stamp = [&](int k) { if (N.clk && tid == 0) N.clk[8 * (size_t)(wg0 + blockIdx.x) + k] = wall_clock64(); };
    stamp(0);
    if (N.abort && *N.abort == N.abort_id) return;                 // (a discarded speculative trial drains)
    if (first && N.x_poll)                                         // (nothing reads xn before the back pass of this solve, launches later)
        for (int i = blockIdx.x * NTH + tid; i < N.n_x3; i += gridDim.x * NTH) reinterpret_cast<unsigned long long*>(N.xn)[i] = ND_POISON;
    // ---- requests first: this thread's original entries (descriptor and values: one round trip) and the Schur complements the
    // children left in this front's assembly slots (dense, in this front's own index space: contiguous 16-byte loads)
    auto entry_row = [&](const NdEnt& E) {                         // W row of an entry's first row, -1: not in this workgroup's blocks
        const int fr_row = 3 * (int)E.r;
        if (fr_row < s) return fr_row;
        if (inv) return -1;
        const int rb = fr_row - s;
        if (rb >= ND_TB * I && rb < ND_TB * I + ND_TB) return rowI0 + rb - ND_TB * I;
        if (two && rb >= ND_TB * J && rb < ND_TB * J + ND_TB) return rowJ0 + rb - ND_TB * J;
        return -1;
    };
    constexpr int NE = 512 / NTH;
    NdEnt En[NE];
    double ev[NE][9];
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int e = F.ent_off + min(tid + NTH * u, F.n_ent - 1);
        En[u] = N.ent[e];
        const double* v = N.ev + 9 * (size_t)e;
#pragma unroll
        for (int a = 0; a < 9; ++a) ev[u][a] = v[a];
    }
    if (inv) { if (tid < 32) pmi[tid] = -1; }
    else if (tid < 16) { const int np = 16 * I + tid; pmi[tid] = np <= F.b / 3 ? N.pmap[F.pmap_off + np] : (int16_t)-1; }
    else if (tid < 32) { const int np = 16 * J + tid - 16; pmj[tid - 16] = np <= F.b / 3 ? N.pmap[F.pmap_off + np] : (int16_t)-1; }
    const size_t slot = (size_t)(m + 1) * F.ldA;
    const double* A0 = N.A + F.A_off;
    nd_v4d acc[NT3];
#pragma unroll
    for (int q = 0; q < NT3; ++q) acc[q] = nd_v4d{0.0, 0.0, 0.0, 0.0};
    if (chained && wd.pad > 0) {                                   // (pad: the tiles this front's children deliver INSIDE this launch, per solve)
        // every level in one launch: wait until the children's workgroups (smaller block indices: dispatched before this one, so a full
        // chip cannot deadlock; bounded all the same) have delivered their tiles -- wd.pad of them per solve -- then read past stale lines
        if (tid == 0) {
            const int want = chained * wd.pad;
            int spins = 0;
            while (__hip_atomic_load(N.fcnt + F.cmap_off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 23)) { N.flags[2] = 2; break; }
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    if (F.n_ch > 0) {                                              // F22 tile (I, J): straight into the accumulators of the matrix cores
        // (the children wrote every element once, at (larger, smaller) of its two positions here: a diagonal tile's upper half is
        // read at its mirror position)
        auto tile_off = [&](int r, int cc) {
            const int fr = s + ND_TB * I + r, fc = s + ND_TB * J + cc;
            return (size_t)max(fr, fc) * F.ldA + min(fr, fc);
        };
        double tv[2][NT3][4];
#pragma unroll
        for (int t3 = 0; t3 < NT3; ++t3) {
            const int t = wave + NW * t3, ti = t / 3, tj = t - 3 * ti;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = 16 * ti + (lane >> 4) + 4 * g, cc = 16 * tj + (lane & 15);
                const bool in = t < 9 && r < rI && cc < cJ && ND_TB * J + cc < F.b;
                const size_t o = in ? tile_off(r, cc) : 0;
                tv[0][t3][g] = A0[o];
                tv[1][t3][g] = A0[(F.n_ch > 1 ? slot : 0) + o];
                if (!in) { tv[0][t3][g] = 0.0; tv[1][t3][g] = 0.0; }
            }
        }
#pragma unroll
        for (int t3 = 0; t3 < NT3; ++t3)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[t3][g] = tv[0][t3][g] + (F.n_ch > 1 ? tv[1][t3][g] : 0.0);
        for (int k = 2; k < F.n_ch; ++k)                           // (more than two children: a separator whose halves fell apart)
#pragma unroll
            for (int t3 = 0; t3 < NT3; ++t3) {
                const int t = wave + NW * t3, ti = t / 3, tj = t - 3 * ti;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int r = 16 * ti + (lane >> 4) + 4 * g, cc = 16 * tj + (lane & 15);
                    if (t < 9 && r < rI && cc < cJ && ND_TB * J + cc < F.b) acc[t3][g] += A0[k * slot + tile_off(r, cc)];
                }
            }
    }
    if (F.n_ch == 0) {                                             // a leaf: the panel starts from zero (and the unit diagonals below)
        double2* W2 = reinterpret_cast<double2*>(W);
        for (int i = tid; i < (nrow * ND_LD) >> 1; i += NTH) W2[i] = make_double2(0.0, 0.0);
        __syncthreads();
        if (tid < s16 - s) W[(s + tid) * ND_LD + s + tid] = 1.0;   // padding columns: unit diagonal
        if (inv && tid < s) W[(s16 + tid) * ND_LD + tid] = 1.0;
    } else {
        // panel rows of this workgroup <- sum of the children's slots: W row wr = ty + RG i is front row fr; thread (tx, ty) takes the column
        // pairs 2 tx + 32 j.  The pass writes EVERY element of the panel (rows < nrow, columns < 96) -- zero where no slot element
        // belongs, one on the unit diagonals of the padding columns and of the inverse workgroup's identity -- so nothing is zeroed
        // first and no barrier stands between these loads and the requests above: one memory round trip for entries, tile and panel
        constexpr int RG = NTH / 16;
        const int tx = tid & 15, ty = tid >> 4;
#pragma unroll 1
        for (int i0 = 0; RG * i0 < nrow; i0 += 6) {
            double2 v0[6][3], v1[6][3];
            bool ok[6][3], ok2[6][3];                               // (second column of the pair: only below the row's limit)
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int wr = ty + RG * (i0 + i);
                int fr = -1, lim = s;                              // columns [0, lim) of front row fr
                if (wr < s16) { if (wr < s) { fr = wr; lim = wr + 1; } }
                else if (wr < s16 + ND_TB) { if (wr - s16 < rI) fr = s + ND_TB * I + wr - s16; }
                else if (two && wr < nrow) fr = s + ND_TB * J + wr - s16 - ND_TB;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int q = 2 * tx + 32 * j;
                    ok[i][j] = fr >= 0 && q < lim;
                    ok2[i][j] = fr >= 0 && q + 1 < lim;
                    const size_t o = ok[i][j] ? (size_t)fr * F.ldA + q : 0;
                    v0[i][j] = *reinterpret_cast<const double2*>(A0 + o);
                    v1[i][j] = *reinterpret_cast<const double2*>(A0 + (F.n_ch > 1 ? slot : 0) + o);
                }
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int wr = ty + RG * (i0 + i);
                if (wr >= nrow) continue;
                // the one of this row: padding columns' unit diagonal (rows s .. s16), the identity under F11 (inverse workgroup)
                const int one = wr >= s && wr < s16 ? wr : (inv && wr >= s16 && wr - s16 < s ? wr - s16 : -1);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int q = 2 * tx + 32 * j;
                    double* d = W + wr * ND_LD + q;
                    d[0] = ok[i][j] ? v0[i][j].x + (F.n_ch > 1 ? v1[i][j].x : 0.0) : (q == one ? 1.0 : 0.0);
                    d[1] = ok2[i][j] ? v0[i][j].y + (F.n_ch > 1 ? v1[i][j].y : 0.0) : (q + 1 == one ? 1.0 : 0.0);
                }
            }
        }
        for (int k = 2; k < F.n_ch; ++k) {
            __syncthreads();
            for (int idx = tid; idx < nrow * 48; idx += NTH) {
                const int wr = idx / 48, q = 2 * (idx - 48 * wr);
                int fr = -1, lim = s;
                if (wr < s16) { if (wr < s) { fr = wr; lim = wr + 1; } }
                else if (wr < s16 + ND_TB) { if (wr - s16 < rI) fr = s + ND_TB * I + wr - s16; }
                else if (two) fr = s + ND_TB * J + wr - s16 - ND_TB;
                if (fr < 0 || q >= lim) continue;
                const double2 v = *reinterpret_cast<const double2*>(A0 + k * slot + (size_t)fr * F.ldA + q);
                W[wr * ND_LD + q] += v.x;
                if (q + 1 < lim) W[wr * ND_LD + q + 1] += v.y;
            }
        }
        __syncthreads();
    }
    stamp(1);
    {
        auto put_entry = [&](const NdEnt& E, const double* v, int wr) {
            const uint32_t kind = E.src >> ND_KIND_SHIFT;
            double* dst = W + (size_t)wr * ND_LD + 3 * (int)E.c;
            if (kind == 2) { dst[0] += v[0]; dst[1] += v[1]; dst[2] += v[2]; return; }
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int j = 0; j < 3; ++j) dst[a * ND_LD + j] += v[3 * a + j] + ((kind == 0 && a == j) ? lam : 0.0);
        };
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int wr = tid + NTH * u < F.n_ent ? entry_row(En[u]) : -1;
            if (wr >= 0) put_entry(En[u], ev[u], wr);
        }
        for (int e = tid + NTH * NE; e < F.n_ent; e += NTH) {      // (fronts with more than 512 entries)
            const NdEnt E = N.ent[F.ent_off + e];
            const int wr = entry_row(E);
            if (wr < 0) continue;
            double t9[9];
            for (int a = 0; a < 9; ++a) t9[a] = N.ev[9 * (size_t)(F.ent_off + e) + a];
            put_entry(E, t9, wr);
        }
    }
    __syncthreads();
    stamp(2);
    // ---- panel factorisation of [F11; F21_I; F21_J] by 16-column steps.  Per step: (A) wave 0 factorises the diagonal block
    // while waves 1..3 apply the PREVIOUS panel to the block columns behind the next one; (B) every thread solves one panel row
    // against the block; (C) the next step's block column is updated by all four waves.  Every piece of code appears once.
    const int nb = s16 >> 4, nrt = nrow >> 4;
    int bad = 0;
    long long tA = 0, tB = 0, tq = 0;                              // (NRS_ND_DBG: time of wave 0 in steps A and B)
    if constexpr (W32) {
        // 32-column steps (round 5): blocks a and b = a + 1 per step.  Wave 0 runs the chain that cannot be shortened -- the diagonal block
        // of a, the sixteen panel rows of block b against it, their product into the diagonal block of b, the diagonal block of b --
        // and the other waves do everything else next to it: (P1) the two panels of the step before into block columns a and b, (P2) the
        // rows below block b against block a, (P3) those rows' product into block column b and the two panels of the step before into
        // the columns behind b, (P4, all waves) the rows against block b.  Four barriers per 32 columns as before, but the chain no longer waits for the rows and their products between its
        // two diagonal blocks.  Every tile sees the same operations in the same order as in the 16-column form: the same bits.
#pragma unroll 1
        for (int a = 0; a < nb; a += 2) {
            const int ka = 16 * a, b = a + 1, kbb = 16 * b;
            const bool pair = b < nb;
            if (N.clk) tq = wall_clock64();
            if (wave == 0) {                                       // P1
                if (a > 0) nd_update<2>(W, lane, ka - 32, a, a + 1, a + 1, 0, 1);
                nd_diag_factor(W, dinv, ka, lane, bad);
            } else if (a > 0) nd_update<2>(W, lane, ka - 32, a, min(a + 2, nb), nrt, wave - 1, NW - 1, 1);   // (block columns a and b only: the columns behind them get theirs in P3, next to the chain's second diagonal block)
            __syncthreads();
            if (N.clk) { const long long t = wall_clock64(); tA += t - tq; tq = t; }
            if (!pair) {                                           // (an odd last block: its rows, and done)
                if (ka + 16 + 64 * wave < nrow) nd_b_row(W, dinv, ka, min(ka + 16 + tid, nrow - 1), ka + 16 + tid < nrow, lane);
                __syncthreads();
                if (N.clk) tB += wall_clock64() - tq;
                break;
            }
            if (wave == 0) {                                       // P2: rows of block b against block a, then their product into (b, b)
                nd_b_row(W, dinv, ka, kbb + (lane & 15), lane < 16, lane);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                nd_update<1>(W, lane, ka, b, b + 1, b + 1, 0, 1);
            } else {
                const int row = kbb + 16 + (tid - 64);
                if (kbb + 16 + 64 * (wave - 1) < nrow) nd_b_row(W, dinv, ka, min(row, nrow - 1), row < nrow, lane);
            }
            __syncthreads();
            if (N.clk) { const long long t = wall_clock64(); tB += t - tq; tq = t; }
            if (wave == 0) nd_diag_factor(W, dinv, kbb, lane, bad);                                   // P3
            else {
                nd_update<1>(W, lane, ka, b, b + 1, nrt, wave - 1, NW - 1, 1);
                if (a > 0) nd_update<2>(W, lane, ka - 32, a + 2, nb, nrt, wave - 1, NW - 1, 0);
            }
            __syncthreads();
            if (N.clk) { const long long t = wall_clock64(); tA += t - tq; tq = t; }
            if (kbb + 16 + 64 * wave < nrow) nd_b_row(W, dinv, kbb, min(kbb + 16 + tid, nrow - 1), kbb + 16 + tid < nrow, lane);   // P4
            __syncthreads();
            if (N.clk) tB += wall_clock64() - tq;
        }
    } else
#pragma unroll 1
    for (int kb = 0; kb < nb; ++kb) {
        const int k0 = 16 * kb;
        if (N.clk) tq = wall_clock64();
        // (wave 0: the previous panel's update of THIS diagonal block, then its factorisation (A); waves 1..3 meanwhile apply the
        // previous panel to everything else right of it -- the rest of this block column included: only B is done by all four)
        if (wave == 0) {
            if (kb > 0) nd_update(W, lane, k0 - 16, kb, kb + 1, kb + 1, 0, 1);
            nd_diag_factor(W, dinv, k0, lane, bad);
        } else if (kb > 0) nd_update(W, lane, k0 - 16, kb, nb, nrt, wave - 1, NW - 1, 1);
        __syncthreads();
        if (N.clk) { const long long t = wall_clock64(); tA += t - tq; tq = t; }
        if (k0 + 16 + 64 * wave < nrow) {                          // (wave-uniform: the waves beyond the panel's rows stay out of the VALU's way)
            // (B) one panel row per thread: x L_kk^T = a, column by column.  L_kk sits in registers, row (lane & 15) in every 16-lane
            // row of the wave, and L[q][p] reaches the FMA through a DPP row broadcast: no LDS read inside the substitution
            // (it was 136 broadcast reads per thread: 1.07 -> 0.4 us per step).  Every lane computes; only the store is predicated.
            const int row = min(k0 + 16 + tid, nrow - 1), li = lane & 15;
            double x[16], lk[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) { x[q] = W[row * ND_LD + k0 + q]; lk[q] = W[(k0 + li) * ND_LD + k0 + q]; }
            double di[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) di[q] = dinv[k0 + q];
            nd_b_cols<0>(x, lk, di);
            if (k0 + 16 + tid < nrow) {
#pragma unroll
                for (int q = 0; q < 16; ++q) W[row * ND_LD + k0 + q] = x[q];
            }
        }
        __syncthreads();
        if (N.clk) tB += wall_clock64() - tq;
    }
    stamp(3);
    if (N.clk && tid == 0) { N.clk[8 * (size_t)(wg0 + blockIdx.x) + 6] = tA; N.clk[8 * (size_t)(wg0 + blockIdx.x) + 7] = tB; }
    // ---- Schur tile: U_IJ = F22_IJ - L21_I L21_J^T (k outermost: the wave's tiles advance together, operands of four k-steps in flight),
    // written into the parent's assembly slot at the parent's positions of its rows and columns (the lower one of the two)
    if (F.par >= 0 && !inv) {
        int ti[NT3], tj[NT3];
#pragma unroll
        for (int t3 = 0; t3 < NT3; ++t3) { const int t = min(wave + NW * t3, 8); ti[t3] = t / 3; tj[t3] = t - 3 * ti[t3]; }
        const bool last = wave + NW * (NT3 - 1) < 9;                // (tiles 0..8 over the waves: wave 0 has one more than the others)
#pragma unroll 1
        for (int k4 = 0; k4 < (s16 >> 4); ++k4) {
            double av[NT3][4], bv[NT3][4];
#pragma unroll
            for (int t3 = 0; t3 < NT3; ++t3)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    av[t3][kk] = -W[(rowI0 + 16 * ti[t3] + (lane & 15)) * ND_LD + 16 * k4 + 4 * kk + (lane >> 4)];
                    bv[t3][kk] = W[(rowJ0 + 16 * tj[t3] + (lane & 15)) * ND_LD + 16 * k4 + 4 * kk + (lane >> 4)];
                }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int t3 = 0; t3 < NT3 - 1; ++t3) acc[t3] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t3][kk], bv[t3][kk], acc[t3], 0, 0, 0);
                if (last) acc[NT3 - 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[NT3 - 1][kk], bv[NT3 - 1][kk], acc[NT3 - 1], 0, 0, 0);
            }
        }
        double* Ap = N.A + F.pA_off;
#pragma unroll
        for (int t3 = 0; t3 < NT3; ++t3) {
            if (wave + NW * t3 >= 9) break;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = 16 * ti[t3] + (lane >> 4) + 4 * g, cc = 16 * tj[t3] + (lane & 15);
                const int Cc = ND_TB * J + cc;
                if (r < rI && cc < cJ && Cc < F.b && (two || r >= cc)) {         // (a diagonal tile: its lower half)
                    const int PR = 3 * (int)pmi[r / 3] + r % 3, PC = 3 * (int)(two ? pmj : pmi)[cc / 3] + cc % 3;
                    Ap[(size_t)max(PR, PC) * F.pldA + min(PR, PC)] = acc[t3][g];  // ONE store per element: the parent reads lower positions only
                }
            }
        }
    }
    stamp(4);
    if (chained && !inv && F.par >= 0) {                           // this tile is in the parent's slot: count it (release: the stores first)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(N.fcnt + F.par, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (inv) {                                                     // (L11^-1)^T, upper triangular, behind the panel
        const int tx = tid & 31, ty = tid >> 5;
        double* LT = N.Lp + F.L_off + (size_t)(m + 2) * s;
        for (int r = ty; r < s; r += NTH / 32)
            for (int q = tx; q < s; q += 32) LT[(size_t)r * s + q] = q >= r ? W[(s16 + r) * ND_LD + q] : 0.0;
    }
    // ---- the factor: block I's rows of L21 (and y^T) by the DIAGONAL workgroups (I, I) -- those run in every form of a level, also when its
    // off-diagonal tiles come from k_nd_tile, which reads these rows back -- L11 and 1 / diag by (0, 0)
    if (J == I && !inv) {
        const int tx = tid & 31, ty = tid >> 5;
        double* L = N.Lp + F.L_off;
        for (int r = ty; r < rI; r += NTH / 32)
            for (int q = tx; q < s; q += 32) L[(size_t)(s + ND_TB * I + r) * s + q] = W[(rowI0 + r) * ND_LD + q];
        if (I == 0) {
            for (int p = ty; p < s; p += NTH / 32)
                for (int q = tx; q < s; q += 32) L[(size_t)p * s + q] = q <= p ? W[p * ND_LD + q] : 0.0;
            if (tid < s) L[(size_t)(m + 1) * s + tid] = dinv[tid];
            if (bad && lane == 0) N.flags[2] = 1;                  // (wave 0 saw the pivots)
        }
    }
    stamp(5);
}

// ---- the off-diagonal Schur tiles of a CROWDED level (more workgroups than CUs) in a launch of their own: U_IJ = F22_IJ - L21_I L21_J^T from the
// rows of L21 the diagonal workgroups (I, I), (J, J) of the launch before left in the factor -- instead of every (I, J) workgroup factorising
// the front's panel again for its one tile (13 us of panel for 4.4 us of tile, three rounds of workgroups at one per CU on the lowest level of
// a 4.4k-point frame).  Same operands, same matrix-core sequence, same accumulation order as k_nd_level's tile: the same bits.  LDS: two
// 48-row blocks (74 KB), two workgroups per CU.
constexpr int ND_TILE_LDS = 2 * ND_TB * ND_LD;                     // doubles
template <int NTH>
__global__ __launch_bounds__(NTH) void k_nd_tile(NdDev N, int wg0) {
    extern __shared__ double sm[];
    constexpr int NW = NTH / 64, NT3 = (9 + NW - 1) / NW;          // waves; tiles (of nine) per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const NdWgD wd = N.wg[wg0 + blockIdx.x];
    if (N.abort && *N.abort == N.abort_id) return;
    const int I = wd.I, J = wd.J;
    const NdFrontD& F = wd.F;
    const int s = F.s, s16 = (s + 15) & ~15, b1 = F.b + 1, m = s + F.b;
    const int rI = min(ND_TB, b1 - ND_TB * I), cJ = ND_TB;          // (J < I: a full block)
    double* LI = sm;
    double* LJ = sm + ND_TB * ND_LD;
    int16_t* pmi = reinterpret_cast<int16_t*>(sm + ND_TILE_LDS);
    int16_t* pmj = pmi + 16;
    auto stamp = [&](int k) { if (N.clk && tid == 0) N.clk[8 * (size_t)(wg0 + blockIdx.x) + k] = wall_clock64(); };
    stamp(0);
    if (tid < 16) { const int np = 16 * I + tid; pmi[tid] = np <= F.b / 3 ? N.pmap[F.pmap_off + np] : (int16_t)-1; }
    else if (tid < 32) { const int np = 16 * J + tid - 16; pmj[tid - 16] = np <= F.b / 3 ? N.pmap[F.pmap_off + np] : (int16_t)-1; }
    // requests first: the two row blocks of the factor (contiguous: rows of s doubles), then the children's slots of this tile
    const double* L = N.Lp + F.L_off;
    const double* srcI = L + (size_t)(s + ND_TB * I) * s;
    const double* srcJ = L + (size_t)(s + ND_TB * J) * s;
    constexpr int NL = (ND_TB * ND_S16 + NTH - 1) / NTH;           // values per thread and block at most (18 on 256 threads)
    double vi[NL], vj[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) {
        const int i = tid + NTH * u;
        vi[u] = i < rI * s ? srcI[i] : 0.0;
        vj[u] = i < cJ * s ? srcJ[i] : 0.0;
    }
    const size_t slot = (size_t)(m + 1) * F.ldA;
    const double* A0 = N.A + F.A_off;
    nd_v4d acc[NT3];
#pragma unroll
    for (int q = 0; q < NT3; ++q) acc[q] = nd_v4d{0.0, 0.0, 0.0, 0.0};
    if (F.n_ch > 0) {                                              // F22 tile (I, J) of the children, as in k_nd_level
        auto tile_off = [&](int r, int cc) {
            const int fr = s + ND_TB * I + r, fc = s + ND_TB * J + cc;
            return (size_t)max(fr, fc) * F.ldA + min(fr, fc);
        };
        double tv[2][NT3][4];
#pragma unroll
        for (int t3 = 0; t3 < NT3; ++t3) {
            const int t = wave + NW * t3, ti = t / 3, tj = t - 3 * ti;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = 16 * ti + (lane >> 4) + 4 * g, cc = 16 * tj + (lane & 15);
                const bool in = t < 9 && r < rI && cc < cJ && ND_TB * J + cc < F.b;
                const size_t o = in ? tile_off(r, cc) : 0;
                tv[0][t3][g] = A0[o];
                tv[1][t3][g] = A0[(F.n_ch > 1 ? slot : 0) + o];
                if (!in) { tv[0][t3][g] = 0.0; tv[1][t3][g] = 0.0; }
            }
        }
#pragma unroll
        for (int t3 = 0; t3 < NT3; ++t3)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[t3][g] = tv[0][t3][g] + (F.n_ch > 1 ? tv[1][t3][g] : 0.0);
        for (int k = 2; k < F.n_ch; ++k)
#pragma unroll
            for (int t3 = 0; t3 < NT3; ++t3) {
                const int t = wave + NW * t3, ti = t / 3, tj = t - 3 * ti;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int r = 16 * ti + (lane >> 4) + 4 * g, cc = 16 * tj + (lane & 15);
                    if (t < 9 && r < rI && cc < cJ && ND_TB * J + cc < F.b) acc[t3][g] += A0[k * slot + tile_off(r, cc)];
                }
            }
    }
    // into LDS at the panel's leading dimension; what the matrix cores read beyond the blocks (columns s .. s16, rows rI .. 48 of a partial
    // block I) is zero.  (row = i / s by a float reciprocal: (i + 0.5) / s is never closer than 0.5 / 96 to an integer)
    const float invs = 1.0f / (float)s;
#pragma unroll
    for (int u = 0; u < NL; ++u) {
        const int i = tid + NTH * u;
        const int r = __float2int_rz(((float)i + 0.5f) * invs), q = i - r * s;
        if (i < rI * s) LI[r * ND_LD + q] = vi[u];
        if (i < cJ * s) LJ[r * ND_LD + q] = vj[u];
    }
    for (int i = tid; i < ND_TB * (s16 - s); i += NTH) {            // pad columns of both blocks
        const int r = i / (s16 - s), q = s + i % (s16 - s);
        LI[r * ND_LD + q] = 0.0; LJ[r * ND_LD + q] = 0.0;
    }
    for (int i = tid; i < (ND_TB - rI) * s; i += NTH) LI[(rI + i / s) * ND_LD + i % s] = 0.0;   // rows below a partial block I
    __syncthreads();
    stamp(1); stamp(2); stamp(3);
    if (F.par >= 0) {
        int ti[NT3], tj[NT3];
#pragma unroll
        for (int t3 = 0; t3 < NT3; ++t3) { const int t = min(wave + NW * t3, 8); ti[t3] = t / 3; tj[t3] = t - 3 * ti[t3]; }
        const bool last = wave + NW * (NT3 - 1) < 9;
#pragma unroll 1
        for (int k4 = 0; k4 < (s16 >> 4); ++k4) {
            double av[NT3][4], bv[NT3][4];
#pragma unroll
            for (int t3 = 0; t3 < NT3; ++t3)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    av[t3][kk] = -LI[(16 * ti[t3] + (lane & 15)) * ND_LD + 16 * k4 + 4 * kk + (lane >> 4)];
                    bv[t3][kk] = LJ[(16 * tj[t3] + (lane & 15)) * ND_LD + 16 * k4 + 4 * kk + (lane >> 4)];
                }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int t3 = 0; t3 < NT3 - 1; ++t3) acc[t3] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t3][kk], bv[t3][kk], acc[t3], 0, 0, 0);
                if (last) acc[NT3 - 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[NT3 - 1][kk], bv[NT3 - 1][kk], acc[NT3 - 1], 0, 0, 0);
            }
        }
        double* Ap = N.A + F.pA_off;
#pragma unroll
        for (int t3 = 0; t3 < NT3; ++t3) {
            if (wave + NW * t3 >= 9) break;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = 16 * ti[t3] + (lane >> 4) + 4 * g, cc = 16 * tj[t3] + (lane & 15);
                const int Cc = ND_TB * J + cc;
                if (r < rI && cc < cJ && Cc < F.b) {
                    const int PR = 3 * (int)pmi[r / 3] + r % 3, PC = 3 * (int)pmj[cc / 3] + cc % 3;
                    Ap[(size_t)max(PR, PC) * F.pldA + min(PR, PC)] = acc[t3][g];
                }
            }
        }
    }
    stamp(4); stamp(5);
}

// back substitution, x_own = L11^-T (y - L21^T x_bnd), all levels in ONE launch: one workgroup per front (roots included: no
// boundary, nothing to wait for), top-down in block order.  A workgroup first brings everything that does not depend on the unknowns above it on chip -- (L11^-1)^T into LDS, L21
// into registers (the first 32 rows per thread group) and LDS (as many further rows as fit), y, output indices.  Its boundary is
// sorted by owner (NdFrontD::seg_off: the parent's unknowns first, the root's last), and the owners finish root first: the
// workgroup takes the segments from the far end, waits for each owner's unknowns (round 5: every thread polls the values it stages --
// agent-scope atomic loads past the caches -- until they are no longer the poison of this solve; NRS_ND_BACK_FLAGS=1: the owner's
// flag, release / acquire at agent scope) and adds that owner's part of L21^T x_bnd -- so whatever does not fit on chip (the tail of a large
// boundary: the oldest ancestors) is read from global memory while the nearer ancestors are still busy, and what is left when the
// parent publishes is its own segment out of registers / LDS, the product with (L11^-1)^T and the publication: ~5 us per level,
// no triangular solve, no global read of the factor on the critical path.  A workgroup only waits for one with a smaller block
// index (dispatched before it), so a full chip cannot deadlock; the wait is bounded all the same.
constexpr int ND_BACK_UR = 32;
__host__ __device__ inline int nd_back_fixed_doubles(int b) { return ND_S16 * ND_LD + 512 + 128 + ((b + 1) & ~1) + ((b / 3 + 2) >> 1); }
__global__ __launch_bounds__(256) void k_nd_back(NdDev N, int clk0, int n_fronts, int epoch, int lds_doubles) {
    extern __shared__ double sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = n_fronts - 1 - (int)blockIdx.x;
    const NdFrontD F = N.lvl_fr[li];                                // (descriptors in level order)
    const int s = F.s, b = F.b, m = s + b;
    if (blockIdx.x == 0 && tid == 0) { N.flags[1] = 1; __threadfence(); N.flags[0] = 1; }   // (read by the host after the launch has completed)
    // (no abort test in this launch: its workgroups wait for each other, and a discarded solve's back pass that lost some of them was measured
    // to fault; it runs to its end on whatever the drained factorisation left -- 60 us, nobody reads the result)
    double* Ls = sm;                                               // (L11^-1)^T, [s][ND_LD]
    double* part = Ls + ND_S16 * ND_LD;                            // [4][128]
    double* tv = part + 512;                                       // [128]: y - L21^T x_bnd
    double* xb = tv + 128;                                         // [b]
    int* bnode = reinterpret_cast<int*>(xb + ((b + 1) & ~1));      // [b / 3]: nodes of the boundary
    double* L21s = xb + ((b + 1) & ~1) + ((b / 3 + 2) >> 1);
    const double* L = N.Lp + F.L_off;
    auto stamp = [&](int k) { if (N.clk && tid == 0) N.clk[8 * (size_t)(clk0 + li) + k] = wall_clock64(); };
    stamp(0);
    // column q of L21 per thread, its rows dealt to 256 / SQ thread groups
    const int SQ = s <= 64 ? 64 : 128, ng = 256 / SQ;
    const int q = tid & (SQ - 1), g = tid / SQ;
    const int nreg = min(b, ng * ND_BACK_UR);                       // rows [0, nreg): registers; [nreg, nreg + nl): LDS; the rest (huge boundaries): global
    const int nl = max(0, min(b - nreg, (lds_doubles - nd_back_fixed_doubles(b)) / s));
    NdOut xo = {};
    if (wave == 0) xo = nd_out_request(N, F, lane);
    const double yq = tid < s ? L[(size_t)m * s + tid] : 0.0;
    for (int i = tid; i < b / 3; i += 256) bnode[i] = N.bnd[F.bnd_off + i];
    const double* Lq = L + (size_t)s * s + min(q, s - 1);
    double lr[ND_BACK_UR];
#pragma unroll
    for (int u = 0; u < ND_BACK_UR; ++u) lr[u] = Lq[(size_t)max(min(g + u * ng, b - 1), 0) * s];     // (a root has no boundary: the value is not used)
    {
        // (all requests of a staging step in flight together: a plain copy loop waits for every load before the next goes out --
        // 36 + 40 dependent round trips, 70 us for a front with a boundary of 70 nodes)
        const int tx = tid & 31, ty = tid >> 5;
        const double* LT = L + (size_t)(m + 2) * s;
        double v[12][3];
#pragma unroll
        for (int i = 0; i < 12; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int r = ty + 8 * i, p = tx + 32 * j;
                v[i][j] = (r < s && p < s && p >= r) ? LT[(size_t)r * s + p] : 0.0;
            }
#pragma unroll
        for (int i = 0; i < 12; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int r = ty + 8 * i, p = tx + 32 * j;
                if (r < s && p < s) Ls[r * ND_LD + p] = v[i][j];
            }
        const double* L2 = L + (size_t)(s + nreg) * s;              // (rows are contiguous; s is a multiple of 3, the panel offset of 2 doubles: 8-byte accesses)
        const int n = nl * s;
#pragma unroll 1
        for (int i0 = tid; i0 < n; i0 += 256 * 20) {
            double w[20];
#pragma unroll
            for (int u = 0; u < 20; ++u) w[u] = L2[min(i0 + 256 * u, n - 1)];
#pragma unroll
            for (int u = 0; u < 20; ++u) if (i0 + 256 * u < n) L21s[i0 + 256 * u] = w[u];
        }
    }
    stamp(5);
    double a8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int2* seg = reinterpret_cast<const int2*>(N.seg) + F.seg_off;
#pragma unroll 1
    for (int sg = F.n_seg - 1; sg >= 0; --sg) {
        const int2 S2 = seg[sg];
        const int r0 = sg > 0 ? seg[sg - 1].y : 0, r1 = S2.y;
        if (N.x_poll) {
            // every thread polls the unknowns it stages until they are there: no flag, no fence -- one memory round trip between an
            // ancestor's store and this front's products instead of three (its fence + flag, this front's poll, then the loads)
            if (sg == 0) stamp(4);
            for (int i = r0 + tid; i < r1; i += 256) {
                const double* src = N.xn + 3 * (size_t)bnode[i / 3] + i % 3;
                double v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while ((unsigned long long)__double_as_longlong(v) == ND_POISON) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 22)) { N.flags[2] = 2; break; }   // (cannot happen: ancestors are dispatched first; never hang the device)
                    v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                xb[i] = v;
            }
            __syncthreads();
        } else {
        if (tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(N.done + S2.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {   // (plain polls: one acquire at the end)
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 23)) { N.flags[2] = 2; break; }  // (cannot happen: ancestors are dispatched first; never hang the device)
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");         // (every thread, behind the barrier, as in k_nd_level: what the ancestor published is visible to all of them)
        if (sg == 0) stamp(4);
        for (int i = r0 + tid; i < r1; i += 256) xb[i] = __hip_atomic_load(N.xn + 3 * (size_t)bnode[i / 3] + i % 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        }
        if (sg == 0) stamp(1);
        if (q < s) {
            if (r0 < nreg) {
#pragma unroll
                for (int u = 0; u < ND_BACK_UR; ++u) { const int r = g + u * ng; if (r >= r0 && r < r1) a8[u & 7] += lr[u] * xb[r]; }
            }
            // this thread's rows in [max(r0, nreg), r1): r = g (mod ng)
            int r = max(r0, nreg);
            r += (g - r % ng + ng) % ng;
            const int e1 = min(r1, nreg + nl);
            for (; r + 3 * ng < e1; r += 4 * ng) {
#pragma unroll
                for (int u = 0; u < 4; ++u) a8[u] += L21s[(r + u * ng - nreg) * s + q] * xb[r + u * ng];
            }
            for (; r < e1; r += ng) a8[0] += L21s[(r - nreg) * s + q] * xb[r];
            for (; r < r1; r += 24 * ng) {                         // (rows beyond the chip: 24 requests in flight per thread)
                double l24[24];
#pragma unroll
                for (int u = 0; u < 24; ++u) l24[u] = Lq[(size_t)min(r + u * ng, b - 1) * s];
#pragma unroll
                for (int u = 0; u < 24; ++u) if (r + u * ng < r1) a8[u & 7] += l24[u] * xb[r + u * ng];
            }
        }
    }
    if (F.n_seg == 0) { stamp(4); stamp(1); }                      // (a root: nothing to wait for)
    {
        const double acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
        if (q < 128) {
            for (int gg = g; gg < 4; gg += ng) part[gg * 128 + q] = gg == g ? acc : 0.0;    // (unused group slots: zero)
        }
    }
    __syncthreads();
    if (tid < s) tv[tid] = yq - ((part[tid] + part[128 + tid]) + (part[256 + tid] + part[384 + tid]));
    __syncthreads();
    stamp(2);
    {
        // x = (L11^-1)^T t: row q2 per thread, the columns split over two thread halves (stride ND_LD: conflict-free)
        const int q2 = tid & 127, h = tid >> 7;
        double a0 = 0, a1 = 0;
        if (q2 < s) {
            const double* row = Ls + q2 * ND_LD;
            int p = h;
            for (; p + 2 < s; p += 4) { a0 += row[p] * tv[p]; a1 += row[p + 2] * tv[p + 2]; }
            for (; p < s; p += 2) a0 += row[p] * tv[p];
        }
        __syncthreads();                                           // (part is reused)
        part[h * 128 + q2] = a0 + a1;
    }
    __syncthreads();
    if (wave == 0) {
        const double x0 = lane < s ? part[lane] + part[128 + lane] : 0.0;          // unknowns 0..63 and 64..127 of the front, two per lane
        const double x1 = lane + 64 < s ? part[lane + 64] + part[128 + lane + 64] : 0.0;
        nd_store_x(N, F, xo, lane, x0, x1);
        if (!N.x_poll) {
            __threadfence();                                       // (this wave wrote the unknowns: its release publishes them)
            if (lane == 0) __hip_atomic_store(N.done + F.cmap_off, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    stamp(3);
}

// ---- host side ------------------------------------------------------------------------------------------------------
struct NdSolver {
    NdPlan plan;
    NdDev dev;
    DevBuf own;                      // everything the kernels read: plan arrays, entry values, assembly areas, L, x ...
    DevBuf* buf = &own;              // ... in the solver's own buffer (the tap) or in the context's (engines: reused from frame to frame)
    std::vector<size_t> lvl_shm_fac;
    std::vector<char> h_stage;       // host image of the plan arrays (one upload)
    int epoch = 0;                   // solves so far (the flags of the single-launch back pass count them)
    int chained = 0;                 // ... of which with the single-launch factorisation (its per-front counters count those)
    int chain_from = 0;              // the levels from here up run as ONE launch (their workgroups are resident at once); n_levels: none
    size_t shm_back_all = 0;
    bool attr_set = false;
    double* d_ev = nullptr;
    const NdEnt* d_ent = nullptr;
    int n_alt = 0;                   // further sets of everything a solve WRITES (factor, assembly areas, unknowns, per-front words): speculative LM trials
    size_t alt_stride = 0;           // ... each this many bytes behind the one before (set before nd_upload; nd_alt_dev)
};

// the device view of solve set j >= 0 of the alternates (same plan and entry values, its own factor storage); the caller points out_rows /
// out_pose / flags at its own vectors
static NdDev nd_alt_dev(const NdSolver& S, int j) {
    NdDev D = S.dev;
    const size_t shift = (size_t)(j + 1) * S.alt_stride;
    D.Lp = reinterpret_cast<double*>(reinterpret_cast<char*>(D.Lp) + shift); D.A = reinterpret_cast<double*>(reinterpret_cast<char*>(D.A) + shift);
    D.xn = reinterpret_cast<double*>(reinterpret_cast<char*>(D.xn) + shift);
    D.done = reinterpret_cast<int*>(reinterpret_cast<char*>(D.done) + shift); D.fcnt = reinterpret_cast<int*>(reinterpret_cast<char*>(D.fcnt) + shift);
    return D;
}

static int nd_upload(nrs_ctx* c, NdSolver& S) {
    const NdPlan& P = S.plan;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += al(bytes); return o; };
    const size_t o_seg = take(4 * std::max<size_t>(2, P.seg.size())), o_own = take(4 * P.own.size()), o_bnd = take(4 * std::max<size_t>(1, P.bnd.size())),
                 o_pm = take(2 * std::max<size_t>(1, P.pmap.size())), o_ent = take(sizeof(NdEnt) * P.ent.size()),
                 o_wg = take(sizeof(NdWgD) * (P.wg.size() / 3)), o_lf = take(sizeof(NdFrontD) * P.lvl_fronts.size()), o_ev = take(72 * P.ent.size() + 64),
                 o_L = take(8 * P.L_doubles), o_A = take(8 * std::max<size_t>(2, P.A_doubles) + 64), o_x = take(24 * (size_t)P.n_nodes), o_fl = take(64), o_dn = take(4 * P.fr.size()), o_fc = take(4 * P.fr.size());
    const size_t off1 = off;                                       // (one set ends here)
    S.alt_stride = off1 - o_L;
    off += (size_t)S.n_alt * S.alt_stride;
    NRS_TRY(c->ensure(*S.buf, off));
    char* base = S.buf->as<char>();
    // the plan's arrays go up in ONE copy from a staging image that lives as long as the solver (the copy is asynchronous)
    S.h_stage.assign(o_ev, 0);
    auto put = [&](size_t o, const void* src, size_t bytes) { if (bytes) memcpy(S.h_stage.data() + o, src, bytes); };
    put(o_seg, P.seg.data(), 4 * P.seg.size());
    put(o_own, P.own.data(), 4 * P.own.size());
    put(o_bnd, P.bnd.data(), 4 * P.bnd.size());
    put(o_pm, P.pmap.data(), 2 * P.pmap.size());
    put(o_ent, P.ent.data(), sizeof(NdEnt) * P.ent.size());
    {
        NdWgD* hw = reinterpret_cast<NdWgD*>(S.h_stage.data() + o_wg);
        NdFrontD* hl = reinterpret_cast<NdFrontD*>(S.h_stage.data() + o_lf);
        // (device copies of the descriptor: cmap_off, the host reference's gather map, holds the front's own index)
        // (pad: how many workgroups write into this front's assembly slots in one factorisation -- its children's (I, J) pairs)
        // the top of the tree in one launch: the highest levels whose workgroups are resident at once (one per CU), when that spares at
        // least one launch; a front's counter then counts the tiles of its children INSIDE that launch (the others are complete before it)
        std::vector<int> need(P.fr.size(), 0), lvl_of(P.fr.size(), 0);
        for (int l = 0; l < P.n_levels; ++l)
            for (int i = P.lvl_ptr[l]; i < P.lvl_ptr[l + 1]; ++i) lvl_of[P.lvl_fronts[i]] = l;
        S.chain_from = P.n_levels;
        while (S.chain_from > 0 && P.lvl_wg_ptr[P.n_levels] - P.lvl_wg_ptr[S.chain_from - 1] <= c->prop.multiProcessorCount) --S.chain_from;
        // Measured with 512-thread workgroups (round 5): the launch boundaries are the cheaper hand-over at every size -- 155 us per factorise +
        // solve against 163 chained at 543 points (everything resident), 201 / 223 at 1013, 480 / 501 at 4446 (top seven levels chained) -- so
        // the chained form is opt-in (NRS_ND_CHAIN=1, read when a plan is uploaded; the tests hold it to the per-level form bit for bit)
        if (P.n_levels - S.chain_from < 2 || !c->env("NRS_ND_CHAIN")) S.chain_from = P.n_levels;
        for (size_t f = 0; f < P.fr.size(); ++f)
            if (P.fr[f].par >= 0 && lvl_of[f] >= S.chain_from) need[P.fr[f].par] += P.fr[f].nR * (P.fr[f].nR + 1) / 2;
        for (size_t w = 0; w < P.wg.size() / 3; ++w) { hw[w] = NdWgD{P.fr[P.wg[3 * w]], P.wg[3 * w + 1], P.wg[3 * w + 2], need[P.wg[3 * w]]}; hw[w].F.cmap_off = P.wg[3 * w]; }
        for (size_t i = 0; i < P.lvl_fronts.size(); ++i) { hl[i] = P.fr[P.lvl_fronts[i]]; hl[i].cmap_off = P.lvl_fronts[i]; }
    }
    NRS_HIP(c, hipMemcpyAsync(base, S.h_stage.data(), o_ev, hipMemcpyHostToDevice, c->stream));
    NdDev& D = S.dev;
    memset(&D, 0, sizeof(D));
    D.seg = reinterpret_cast<const int*>(base + o_seg);
    D.own = reinterpret_cast<const int*>(base + o_own); D.bnd = reinterpret_cast<const int*>(base + o_bnd);
    D.pmap = reinterpret_cast<const int16_t*>(base + o_pm); D.ent = reinterpret_cast<const NdEnt*>(base + o_ent);
    D.wg = reinterpret_cast<const NdWgD*>(base + o_wg); D.lvl_fr = reinterpret_cast<const NdFrontD*>(base + o_lf);
    S.d_ev = reinterpret_cast<double*>(base + o_ev); S.d_ent = D.ent;
    D.ev = S.d_ev;
    D.Lp = reinterpret_cast<double*>(base + o_L); D.A = reinterpret_cast<double*>(base + o_A); D.xn = reinterpret_cast<double*>(base + o_x);
    D.flags = reinterpret_cast<int*>(base + o_fl); D.done = reinterpret_cast<int*>(base + o_dn); D.fcnt = reinterpret_cast<int*>(base + o_fc);
    D.n_x3 = 3 * P.n_nodes; D.x_poll = c->env("NRS_ND_BACK_FLAGS") ? 0 : 1;
    S.epoch = 0; S.chained = 0;
    for (int j = 0; j <= S.n_alt; ++j) {
        char* bj = base + (size_t)j * S.alt_stride;
        NRS_HIP(c, hipMemsetAsync(bj + o_fl, 0, off1 - o_fl, c->stream));          // (status words and the fronts' flags)
        // the assembly areas are zero wherever no child ever writes (the written pattern is the same in every factorisation)
        NRS_HIP(c, hipMemsetAsync(bj + o_A, 0, 8 * std::max<size_t>(2, P.A_doubles) + 64, c->stream));
    }
    // dynamic LDS per level: the largest panel / boundary of its fronts
    S.lvl_shm_fac.assign(P.n_levels, 0); S.shm_back_all = 8 * (size_t)nd_back_fixed_doubles(0);
    for (int l = 0; l < P.n_levels; ++l)
        for (int i = P.lvl_ptr[l]; i < P.lvl_ptr[l + 1]; ++i) {
            const NdFrontD& F = P.fr[P.lvl_fronts[i]];
            const int s16 = (F.s + 15) & ~15, nrow = s16 + ND_TB + (F.nR > 1 ? ND_TB : 0);
            S.lvl_shm_fac[l] = std::max(S.lvl_shm_fac[l], sizeof(double) * ((size_t)nrow * ND_LD + ND_S16 + 256) + 2 * 32);
            S.lvl_shm_fac[l] = std::max(S.lvl_shm_fac[l], sizeof(double) * ((size_t)2 * s16 * ND_LD + ND_S16 + 256) + 2 * 32);          // (the inverse workgroup)
            {
                const int ngb = F.s <= 64 ? 4 : 2;
                const size_t want = sizeof(double) * ((size_t)nd_back_fixed_doubles(F.b) + (size_t)std::max(0, F.b - ngb * ND_BACK_UR) * F.s);
                S.shm_back_all = std::max(S.shm_back_all, std::min(want, (size_t)160 * 1024));
                if (sizeof(double) * (size_t)nd_back_fixed_doubles(F.b) > 160 * 1024) return c->fail(NRS_ERR_INVALID, "direct solve: a front's boundary does not fit the LDS");
            }
        }
    if (!S.attr_set) {
        NRS_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_nd_level<256, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        NRS_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_nd_level<512, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        NRS_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_nd_level<256, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        NRS_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_nd_level<512, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        NRS_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_nd_back), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        NRS_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_nd_tile<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        NRS_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_nd_tile<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        S.attr_set = true;
    }
    for (int l = 0; l < P.n_levels; ++l)
        if (S.lvl_shm_fac[l] > 160 * 1024) return c->fail(NRS_ERR_INVALID, "direct solve: a front does not fit the LDS");
    return NRS_OK;
}

// factorise (H + lam I) and solve: 2 x levels launches on the context's stream, no host synchronisation
// back_wait / back_record (speculative trials, several solves in flight on different streams): the back pass is the one launch whose workgroups
// wait for each other, which is safe for ONE such launch at a time -- its lowest unfinished workgroup is always resident or next in its XCD's
// queue -- and not for two: each can fill the CUs of an XCD with waiting workgroups while the workgroup the other's wait for sits in that XCD's
// queue behind them (measured: a 2 s stall ended by the spin bound, NRS_ERR_HIP).  So the back passes of a batch run one after the other:
// this one starts behind the event back_wait and records back_record.  The factorisation's launches wait for nobody and overlap freely.
static int nd_solve_enqueue(nrs_ctx* c, NdSolver& S, double lam, const NdDev* alt = nullptr, int* solve_id = nullptr, hipEvent_t back_wait = nullptr,
                            hipEvent_t back_record = nullptr) {   // alt: the arrays of another solve set (nd_alt_dev)
    const NdPlan& P = S.plan;
    NdDev dev = alt ? *alt : S.dev;
    const int epoch = ++S.epoch;
    dev.abort_id = epoch;
    if (solve_id) *solve_id = epoch;
    // One launch per level.  Opt-in (NRS_ND_CHAIN=1): the TOP of the tree in ONE launch, a front's workgroups waiting for the tiles of its
    // children inside that launch -- the highest levels whose workgroups are all resident at once (one per CU), the whole factorisation
    // for frames of <= ~800 points.  It paid with 256-thread workgroups at 543 points (176 -> 165 us per factorise + solve, round 4) and
    // does not with 512-thread ones (nd_upload); NRS_ND_LEVELS=1, and the phase clocks, put every level in a launch of its own regardless.
    // "Resident at once" is what makes the waits safe and is a property of the device, not a constant: one workgroup per CU (a panel
    // fills most of a CU's LDS), so the bound is the CU count of THIS device (256 on a whole MI355X, fewer in a partition mode).  The
    // no-deadlock argument: a front's workgroups wait only for workgroups of its children, which sit at SMALLER block indices, and the
    // dispatcher hands workgroups of a launch out in block-index order -- observed on every CDNA part, not promised by HIP; hence the
    // bounded spins in k_nd_level / k_nd_back (a wait that runs out raises flags[2] = 2 -> NRS_ERR_HIP, never a hang) and the
    // resident-at-once condition, under which the order does not matter at all.
    const bool per_level = c->env("NRS_ND_LEVELS") != nullptr;     // (read per call: the tests switch it between solves)
    // 512 threads per workgroup unless NRS_ND_THREADS=256 (a level is one workgroup's latency: eight waves shorten its trailing updates,
    // its reads of the children's slots and its Schur tiles; same bits either way)
    const char* nth_env = c->env("NRS_ND_THREADS");
    const bool wide = !(nth_env && atoi(nth_env) == 256);
    // 32-column panel steps (k_nd_level<.., true>) unless NRS_ND_STEP32=0; same bits as the 16-column form
    const char* s32_env = c->env("NRS_ND_STEP32");
    const bool step32 = !(s32_env && atoi(s32_env) == 0);
    int first = 1;                                                 // (the first launch of the solve poisons xn)
    auto level = [&](int n, size_t shm, int wg0, int chained) {
        if (wide && step32) hipLaunchKernelGGL((k_nd_level<512, true>), dim3(n), dim3(512), shm, c->stream, dev, wg0, lam, epoch, chained, first);
        else if (wide) hipLaunchKernelGGL((k_nd_level<512, false>), dim3(n), dim3(512), shm, c->stream, dev, wg0, lam, epoch, chained, first);
        else if (step32) hipLaunchKernelGGL((k_nd_level<256, true>), dim3(n), dim3(256), shm, c->stream, dev, wg0, lam, epoch, chained, first);
        else hipLaunchKernelGGL((k_nd_level<256, false>), dim3(n), dim3(256), shm, c->stream, dev, wg0, lam, epoch, chained, first);
        first = 0;
    };
    const int chain_from = per_level || dev.clk || alt ? P.n_levels : S.chain_from;   // (the per-front counters of the chained form count one set's solves)
    {
        // a CROWDED level (more workgroups than CUs: they would run in rounds, one per CU, each factorising its front's panel for one
        // tile) runs as two launches: the diagonal and inverse workgroups factorise and leave their rows of L21, k_nd_tile makes the
        // off-diagonal tiles from them (NRS_ND_NO_SPLIT=1: one launch per level throughout; the bits are the same)
        const bool no_split = c->env("NRS_ND_NO_SPLIT") != nullptr;
        for (int l = 0; l < chain_from; ++l) {
            const int n = P.lvl_wg_ptr[l + 1] - P.lvl_wg_ptr[l], nA = P.lvl_wg_split[l] - P.lvl_wg_ptr[l];
            if (!no_split && n > c->prop.multiProcessorCount && n > nA) {
                level(nA, S.lvl_shm_fac[l], P.lvl_wg_ptr[l], 0);
                if (wide) hipLaunchKernelGGL(k_nd_tile<512>, dim3(n - nA), dim3(512), sizeof(double) * ND_TILE_LDS + 64, c->stream, dev, P.lvl_wg_split[l]);
                else hipLaunchKernelGGL(k_nd_tile<256>, dim3(n - nA), dim3(256), sizeof(double) * ND_TILE_LDS + 64, c->stream, dev, P.lvl_wg_split[l]);
            } else level(n, S.lvl_shm_fac[l], P.lvl_wg_ptr[l], 0);
        }
        if (chain_from < P.n_levels) {                             // the levels above in one launch (all of them when the whole factorisation is resident at once)
            size_t shm = 0;
            for (int l = chain_from; l < P.n_levels; ++l) shm = std::max(shm, S.lvl_shm_fac[l]);
            level(P.lvl_wg_ptr[P.n_levels] - P.lvl_wg_ptr[chain_from], shm, P.lvl_wg_ptr[chain_from], ++S.chained);
        }
    }
    // (Measured and dropped: the back pass on a second stream next to the last factorisation level -- only roots live there -- so that
    // its workgroups stage their factors while the root is busy.  The two event waits cost more than the ~10 us of staging they hide:
    // 224 -> 245 us per solve at 543 points, 503 -> 525 at 2220.)
    if (back_wait) NRS_HIP(c, hipStreamWaitEvent(c->stream, back_wait, 0));
    hipLaunchKernelGGL(k_nd_back, dim3(P.n_fronts), dim3(256), S.shm_back_all, c->stream, dev, (int)P.wg.size() / 3, P.n_fronts, epoch, (int)(S.shm_back_all / 8));
    NRS_HIP(c, hipGetLastError());
    if (back_record) NRS_HIP(c, hipEventRecord(back_record, c->stream));
    return NRS_OK;
}

// leaf size of the dissection (nodes): ND_LEAFN unless NRS_ND_LEAF says otherwise (a tuning knob: part of the plan cache's key)
static int nd_leaf_n(const nrs_ctx* c) {
    if (const char* v = c->env("NRS_ND_LEAF")) return std::max(4, std::min(ND_LEAFN, atoi(v)));
    return ND_LEAFN;
}
void nd_orient_pairs(const NdPlan& P, const int32_t* pairs, const double* Vp, std::vector<double>& out);   // nrs_host_build.cpp
void nd_stats(const NdPlan& P, int64_t* stats);

// include/nrs.h nrs_debug_nd_solve
int engine_nd_debug_solve(nrs_ctx* c, int n_nodes, const double* pos, const uint8_t* last, int n_pairs, const int* pairs, const double* Dn, const double* Vp,
                          const double* bn, double lam, int repeats, double* x, int64_t* stats, double* ms_per_solve) {
    NRS_HIP(c, hipSetDevice(c->device));
    NdSolver S;
    std::string err;
    if (!nd_build_plan(n_nodes, pos, last, n_pairs, pairs, S.plan, &err, nd_leaf_n(c), ND_SMAXN, true, 0, c->env("NRS_ND_NO_COVER") == nullptr)) return c->fail(NRS_ERR_INVALID, "direct solve: %s", err.c_str());
    nd_stats(S.plan, stats);
    struct Rel { nrs_ctx* c; NdSolver* s; ~Rel() { (void)hipStreamSynchronize(c->stream); c->release(s->own); } } rel{c, &S};
    NRS_TRY(nd_upload(c, S));
    std::vector<double> V, ev(9 * S.plan.ent.size(), 0.0);
    nd_orient_pairs(S.plan, pairs, Vp, V);
    for (size_t e = 0; e < S.plan.ent.size(); ++e) {                // the blocks in entry order (what k_nd_values writes for an engine)
        const uint32_t kind = S.plan.ent[e].src >> ND_KIND_SHIFT, src = S.plan.ent[e].src & ND_SRC_MASK;
        const double* v = kind == 0 ? Dn + 9 * (size_t)src : kind == 1 ? V.data() + 9 * (size_t)src : bn + 3 * (size_t)src;
        for (int a = 0; a < (kind == 2 ? 3 : 9); ++a) ev[9 * e + a] = v[a];
    }
    NRS_HIP(c, hipMemcpyAsync(S.d_ev, ev.data(), 8 * ev.size(), hipMemcpyHostToDevice, c->stream));
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
    if (c->env("NRS_ND_DBG")) {                                    // phase clocks of one solve: mean / max over the workgroups of every launch
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
            double mean[5] = {0, 0, 0, 0, 0}, mx[5] = {0, 0, 0, 0, 0}, mA = 0, mB = 0;
            long long lo = LLONG_MAX, hi = 0;
            const int a = P.lvl_wg_ptr[l], b2 = P.lvl_wg_ptr[l + 1];
            for (int w = a; w < b2; ++w) {
                const long long* q = &h[8 * (size_t)w];
                mA += (double)q[6] / 100.0 / (b2 - a); mB += (double)q[7] / 100.0 / (b2 - a);
                for (int k = 0; k < 5; ++k) { const double d = (double)(q[k + 1] - q[k]) / 100.0; mean[k] += d / (b2 - a); mx[k] = std::max(mx[k], d); }
                lo = std::min(lo, q[0]); hi = std::max(hi, q[5]);
            }
            fprintf(stderr, "[nrs] nd level %2d: %4d wg, span %6.1f us (from %7.1f) | mean / max us: entries %.1f/%.1f gather %.1f/%.1f factor %.1f/%.1f (A %.1f B %.1f) schur %.1f/%.1f store %.1f/%.1f\n", l, b2 - a,
                    (double)(hi - lo) / 100.0, (double)(lo - t00) / 100.0, mean[0], mx[0], mean[1], mx[1], mean[2], mx[2], mA, mB, mean[3], mx[3], mean[4], mx[4]);
        }
        for (int l = P.n_levels - 1; l >= 0; --l) {
            double mean[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
            long long lo = LLONG_MAX, hi = 0;
            const int a = P.lvl_ptr[l], b2 = P.lvl_ptr[l + 1];
            for (int w = a; w < b2; ++w) {
                const long long* q = &h[8 * (P.wg.size() / 3 + (size_t)w)];
                if (q[3] == 0) continue;                            // (a root)
                if (b2 - a <= 2 && c->env("NRS_ND_DBG2")) fprintf(stderr, "   front %d (s %d b %d nseg %d): start %.1f seg-loop-begin %.1f released %.1f gathered %.1f gemv %.1f end %.1f\n", w, P.fr[P.lvl_fronts[w]].s, P.fr[P.lvl_fronts[w]].b, P.fr[P.lvl_fronts[w]].n_seg, (q[0]-t00)/100.0, (q[5]-t00)/100.0, (q[4]-t00)/100.0, (q[1]-t00)/100.0, (q[2]-t00)/100.0, (q[3]-t00)/100.0);
                // (single launch: [4] = released by the parent; "loads" is then the gather of the boundary values only)
                for (int k = 0; k < 3; ++k) { const double d = (double)(q[k + 1] - (k == 0 && q[4] ? q[4] : q[k])) / 100.0; mean[k] += d / (b2 - a); mx[k] = std::max(mx[k], d); }
                lo = std::min(lo, q[4] ? q[4] : q[0]); hi = std::max(hi, q[3]);
            }
            if (hi == 0) continue;
            fprintf(stderr, "[nrs] nd back  %2d: %4d wg, span %6.1f us (from %7.1f) | mean / max us: loads %.1f/%.1f gemv %.1f/%.1f solve %.1f/%.1f\n", l, b2 - a,
                    (double)(hi - lo) / 100.0, (double)(lo - t00) / 100.0, mean[0], mx[0], mean[1], mx[1], mean[2], mx[2]);
        }
    }
    int fl[4] = {0, 0, 0, 0};
    NRS_HIP(c, hipMemcpyAsync(fl, S.dev.flags, sizeof(fl), hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipMemcpyAsync(x, S.dev.xn, 24 * (size_t)n_nodes, hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    if (!fl[0]) return c->fail(NRS_ERR_HIP, "direct solve: the last level did not report completion");
    if (fl[2] == 2) return c->fail(NRS_ERR_HIP, "direct solve: a wait for another workgroup's result timed out");
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
    const NdEnt* ent;                // the plan's original entries; ev: 9 doubles each
    double* ev;
    int n_ent;
    // embedded mode: per entry the skinned observations that add to it (fixed order) with their weight products
    const int* ske_ptr; const int* ske_pt; const double* ske_coef;
};

__device__ inline int nd_hpp_idx(int r, int cc) { return r * 6 - (r * (r - 1)) / 2 + (cc - r); }   // H_pp packed upper-triangular (r <= cc)

// one thread per original entry of the plan: its 3 x 3 block (or its 3 right-hand-side values) of the current linearisation.
// SK (embedded mode): ND_SKL lanes per entry -- all of them form the entry's own part (same addresses: one fetch), each adds up
// every ND_SKL-th skinned observation of the entry's list (a node is reached by ~N * 11 / M observations: ~100 at 5k x 500, a
// serial chain of dependent fetches for one thread) and the partial sums meet in a fixed butterfly: bit-reproducible
constexpr int ND_SKL = 8;
template <bool SK>
__global__ __launch_bounds__(256) void k_nd_values(Dev P, NdVals V) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    const int i = SK ? tid / ND_SKL : tid, sub = SK ? tid % ND_SKL : 0;
    if (i >= V.n_ent) return;
    const NdEnt E = V.ent[i];
    const uint32_t kind = E.src >> ND_KIND_SHIFT, idx = E.src & ND_SRC_MASK;
    double o[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int form = 0;                    // how the skinned observations add to this entry: 0 symmetric block (A), 1 gradient (c), 2 pose-row block (B_p), 3 none
    int half = 0;
    if (kind != 1) {
        const int row = V.node_row[idx];
        form = kind == 2 ? 1 : 0;
        if (row >= 0) {
            if (kind == 2) { o[0] = P.bl[3 * (size_t)row]; o[1] = P.bl[3 * (size_t)row + 1]; o[2] = P.bl[3 * (size_t)row + 2]; }
            else {
                const double* D = P.D + 6 * (size_t)row;
                o[0] = D[0]; o[1] = D[1]; o[2] = D[2]; o[3] = D[1]; o[4] = D[3]; o[5] = D[4]; o[6] = D[2]; o[7] = D[4]; o[8] = D[5];
            }
        } else {
            const int h = -1 - row;                                // half of the pose block (pose 0)
            form = 3;
            if (kind == 2) {
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    o[a] = P.bp[3 * h + a];
                    for (int b = 0; b < P.sk_nblk; ++b) o[a] += P.sk_part[(size_t)b * 32 + 21 + 3 * h + a];
                }
            } else {
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        const int k = nd_hpp_idx(3 * h + min(a, b), 3 * h + max(a, b));
                        o[3 * a + b] = P.Hpp[k];
                        for (int q = 0; q < P.sk_nblk; ++q) o[3 * a + b] += P.sk_part[(size_t)q * 32 + k];
                    }
            }
        }
    } else {
        const NdPairD q = V.pair[idx];
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
                for (int b = 0; b < 3; ++b) o[3 * a + b] = -(qc * v[a] * v[b] + (a == b ? sd : 0.0));     // (symmetric: either orientation)
        } else if (q.kind == 1) {
            // H_{pose half, row} = J_p^T w J_l, J_l = -J R, J_p = -J [-[X_c]x | I] (reprojection_error_with_deformation.cc:52-68), as row_factored() forms them;
            // the pose is eliminated last, so the block's rows are the pose half's components
            form = 2; half = q.a;
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
            form = 3;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {                       // rows: the second half of the pose block
                    const int k = nd_hpp_idx(b, 3 + a);
                    o[3 * a + b] = P.Hpp[k];
                    for (int q2 = 0; q2 < P.sk_nblk; ++q2) o[3 * a + b] += P.sk_part[(size_t)q2 * 32 + k];
                }
        }
    }
    if (SK) {                                                      // the skinned observations that reach this entry: every ND_SKL-th, in list order
        double p[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        const int t1 = form != 3 ? V.ske_ptr[i + 1] : 0;
        for (int t = form != 3 ? V.ske_ptr[i] + sub : 0; t < t1; t += ND_SKL) {
            const double* rec = P.sk_rec + 27 * (size_t)V.ske_pt[t];
            const double cf = V.ske_coef[t];
            if (form == 0) {
                p[0] += cf * rec[0]; p[1] += cf * rec[1]; p[2] += cf * rec[2]; p[3] += cf * rec[1]; p[4] += cf * rec[3]; p[5] += cf * rec[4];
                p[6] += cf * rec[2]; p[7] += cf * rec[4]; p[8] += cf * rec[5];
            } else if (form == 1) { p[0] += cf * rec[6]; p[1] += cf * rec[7]; p[2] += cf * rec[8]; }
            else {
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) p[3 * a + b] += cf * rec[9 + 3 * (3 * half + a) + b];
            }
        }
#pragma unroll
        for (int off = 1; off < ND_SKL; off <<= 1)                  // (x + y on both partners: every lane ends with the same bits)
#pragma unroll
            for (int a = 0; a < 9; ++a) p[a] += __shfl_xor(p[a], off, 64);
#pragma unroll
        for (int a = 0; a < 9; ++a) o[a] += p[a];
        if (sub != 0) return;
    }
    double* out = V.ev + 9 * (size_t)i;
#pragma unroll
    for (int a = 0; a < 9; ++a) out[a] = o[a];
    if (P.sk_n > 0 && kind == 2) {
        // the gradient with the skinned observations' part goes back into the engine's vectors: computeScale = x . (lambda x + b)
        // (optimization_algorithm_levenberg.cpp:167-174, k_apply) is over the whole b
        const int row = V.node_row[idx];
        double* g = row >= 0 ? P.bl + 3 * (size_t)row : P.bp + 3 * (-1 - row);
        g[0] = o[0]; g[1] = o[1]; g[2] = o[2];
    }
    if (P.sk_n > 0 && kind == 0)                                   // lambda_0 = 1e-5 max |diag H| (optimization_algorithm_levenberg.cpp:153-165) sees the added blocks
        atomicMax(reinterpret_cast<unsigned long long*>(P.sk_maxdiag), (unsigned long long)__double_as_longlong(fmax(fabs(o[0]), fmax(fabs(o[4]), fabs(o[8])))));
}

// what a plan's key determines besides the plan itself: the node pairs with their edges and, in the embedded mode, which
// observations add to which plan entry.  Built once per plan; a frame that reuses the plan only fills in its own weights.
struct NdStruct {
    std::vector<int> pairs;                                        // node pairs: row-row couplings (sorted, unique), then the pose's
    std::vector<uint8_t> pkind;                                    // 0 row-row, 1 pose half (first node) - row, 2 pose - pose
    std::vector<int> eptr, eid;                                    // row-row pair -> its edges in edge order: (index << 1) | (0 spring, 1 damper)
    // embedded mode, per plan entry: the observations that add to it (ske_pt) and their coefficient as a product of skinning weights,
    // w[ske_ia] * w[ske_ib] (ske_ib < 0: w[ske_ia] alone); indices into the frame's SK_MAX-wide weight table
    std::vector<int> ske_ptr, ske_pt, ske_ia, ske_ib;
};
// One symbolic factorisation with everything the device needs for it (plan arrays, value descriptors, assembly areas, factor
// storage).  The context keeps the last few (nd_cache): a frame whose optimised set, edges and fixed flags equal an earlier frame's
// -- tracking in steady state: points are lost and edges added every few frames, not every frame -- takes the slot as it is, no
// plan build (1.0 ms at 1k points, 5 ms at 4.5k) and no upload.  The key is the complete input of nd_engine_setup except the
// positions the dissection bisects (they only steer its quality), compared byte for byte.
struct NdSlot {
    NdSolver S;
    std::shared_ptr<NdStruct> st;
    NdVals vals;
    DevBuf ws, vb;                   // S.buf = &ws (plan + factor storage), value descriptors
    std::vector<uint8_t> key;
    std::vector<char> h_vals;        // host image of the value descriptors (one upload)
    uint64_t hash = 0, used = 0;     // (used: LRU stamp)
    bool busy = false, cached = false;
    int n_free = 0, n_pairs = 0;
};
struct NdCache { std::vector<NdSlot*> slots; uint64_t clock = 0, hits = 0, misses = 0; };
constexpr int ND_CACHE_SLOTS = 4;

struct NdEngine {
    NdSlot* slot = nullptr;
    NdSolver& S() { return slot->S; }
    std::vector<double> pos;         // vertex positions the dissection bisects (M x 3, the caller's vertex order)
    std::vector<uint8_t> sig;        // RF_FIXED of every vertex + the pose's flag at set-up
    bool on = false;
};

static void nd_slot_free(nrs_ctx* c, NdSlot* sl) {
    if (!sl) return;
    c->release(sl->ws); c->release(sl->vb);
    delete sl;
}
static void plan_worker_free(nrs_ctx* c);
void nd_cache_free(nrs_ctx* c) {
    plan_worker_free(c);
    NdCache* nc = static_cast<NdCache*>(c->nd_cache);
    if (!nc) return;
    for (NdSlot* sl : nc->slots) nd_slot_free(c, sl);
    delete nc;
    c->nd_cache = nullptr;
}
void nd_cache_stats(nrs_ctx* c, int64_t out[2]) {
    const NdCache* nc = static_cast<const NdCache*>(c->nd_cache);
    out[0] = nc ? (int64_t)nc->hits : 0; out[1] = nc ? (int64_t)nc->misses : 0;
}
static void nd_slot_release(nrs_ctx* c, NdEngine* nd) {            // the engine lets go of its slot: cached ones stay for later frames
    if (!nd || !nd->slot) return;
    if (nd->slot->cached) nd->slot->busy = false; else nd_slot_free(c, nd->slot);
    nd->slot = nullptr; nd->on = false;
}
static inline uint64_t nd_hash(const uint8_t* p, size_t n) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ n;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, p + i, 8); h = (h ^ w) * 0xFF51AFD7ED558CCDull; h ^= h >> 29; }
    for (; i < n; ++i) h = (h ^ p[i]) * 0x100000001B3ull;
    return h;
}

// nrs_options.direct_solve (0: by size, 1: whenever possible, 2: never); NRS_ND / NRS_ND_MAX_ROWS override it for experiments.
// Measured (tools/nd_crossover.py, a2 per frame with a fresh plan per call, direct / PCG ms; flat kNN-16 graph): 129 points 7.9 / 34.6,
// 543: 13.1 / 52.8, 1013: 17.3 / 28.3, 2220: 32.7 / 45.6, 3165: 40.7 / 60.3, 4525: 58.3 / 75.1; on the all-pairs graph (22 neighbours
// per point instead of 13: heavier fronts) 543: 15.5 / 51.7, 1013: 22.4 / 61.8, 2220: 28.3 / 59.7, 3165: 39.8 / 66.3, 4525: 76.3 / 85.0.
// Ahead at every measured size; the default window ends where nothing has been measured.
static bool nd_mode_allows(nrs_ctx* c, int n_free) {
    int mode = c->opt.direct_solve;
    if (const char* ev = c->env("NRS_ND")) mode = atoi(ev) ? 1 : 2;
    const int nmax = c->env("NRS_ND_MAX_ROWS") ? atoi(c->env("NRS_ND_MAX_ROWS")) : 8000;
    return mode != 2 && n_free > 0 && (mode == 1 || n_free <= nmax);
}
static bool nd_wanted(nrs_ctx* c, const Dev& d, int n_free) {
    return nd_mode_allows(c, n_free) && d.K == 1 && d.use_lds && !d.dform && !d.sh_on;
}

// ---- set-up in two phases.  Phase A (nd_prep_run) needs the problem's STRUCTURE only -- which vertices are free, which pairs of
// them an edge or a skinned observation couples -- in the caller's vertex order: it builds the pair lists and the cache key, looks
// the key up and, on a miss, builds the plan.  engine_create runs it on a helper thread next to its own packing of the incidence
// streams (both are host work: a 1k-point frame's 0.8 ms plan build disappears behind the 0.9 ms of packing).  Phase B
// (nd_engine_finish) ties the plan to the engine's row layout: value descriptors (rows, incidence slots), uploads, the slot.
struct NdIn {
    int M = 0;
    const uint8_t* rflag = nullptr;      // M, RF_* bits
    bool pose_fixed = false;
    int n_sp = 0; const int* sp_ij = nullptr;
    int n_dm = 0; const int* dm_idx = nullptr;
    int n_skin = 0; const int* sk_vert = nullptr; const double* sk_om = nullptr;
    const double* vpos = nullptr;        // M x 3: where the dissection bisects
};
struct NdSkT { uint64_t k; int ia, ib; };                       // (pair key, the two weights' places in the weight table)
struct NdPrep {
    bool wanted = false, plan_ok = false;
    std::string err;
    int n_free = 0, n_nodes = 0;
    bool pose_free = false;
    std::vector<int> node_of, node_vtx;                            // vertex -> node (-1: fixed), node -> vertex
    std::shared_ptr<NdStruct> st;                                  // this engine's, or the reused slot's
    std::vector<uint8_t> last;
    std::vector<NdSkT> skt;                                        // embedded mode: (pair, weight places), sorted by pair
    std::vector<int> nl_ptr, nl_ix, pair_sk0, pair_sk1;            // per free node: the weight-table places of the observations that reach it
    std::vector<double> ske_cf;                                    // this frame's coefficients for st->ske_*
    std::vector<uint8_t> key;
    uint64_t hash = 0;
    NdSlot* hit = nullptr;
    NdPlan plan;
    struct PlanWorker* worker = nullptr;                           // the context's helper thread is running nd_prep_run on this object (joined by wait())
    void wait();
    ~NdPrep() { wait(); }
};

// The context's helper thread for the symbolic phase (nd_prep_run next to engine_create's packing).  ONE thread for the context's lifetime, not one
// per frame: a fresh thread starts on a fresh malloc arena, and the ~15 MB of vectors a frame's plan builds were page-faulted in again every frame
// (embedded 500-node frames: pair lists 1.9 ms on a new thread, the whole phase faster inline on the caller's warm heap than next to it on a cold one).
struct PlanWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, busy = false, quit = false;
    bool start() {
        try { th = std::thread([this] { loop(); }); } catch (const std::system_error&) { return false; }
        return true;
    }
    void loop() {
        std::unique_lock<std::mutex> lk(m);
        while (true) {
            cv.wait(lk, [this] { return has_job || quit; });
            if (quit) return;
            std::function<void()> f = std::move(job);
            has_job = false;
            lk.unlock();
            f();                                                   // (nd_prep_run: lets nothing escape)
            lk.lock();
            busy = false;
            cv.notify_all();
        }
    }
    void submit(std::function<void()> f) {
        { std::lock_guard<std::mutex> lk(m); job = std::move(f); has_job = true; busy = true; }
        cv.notify_all();
    }
    void wait() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [this] { return !busy; }); }
    ~PlanWorker() {
        { std::lock_guard<std::mutex> lk(m); quit = true; }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};
inline void NdPrep::wait() { if (worker) { worker->wait(); worker = nullptr; } }
static void plan_worker_free(nrs_ctx* c) {                         // (idle: every NdPrep waits for it before it goes away)
    delete static_cast<PlanWorker*>(c->plan_worker);
    c->plan_worker = nullptr;
}


// records with a key k = (low node << 32) | high node into key order, equal keys in their order of arrival: two stable counting
// passes (least significant half first) -- a comparison sort of the ~10^5 records of a 4.5k-point frame took longer than its plan
template <class T>
static void nd_sort_by_pair(std::vector<T>& v, int n_free) {
    std::vector<T> tmp(v.size());
    std::vector<int> cnt(n_free + 1);
    for (int pass = 0; pass < 2; ++pass) {
        const std::vector<T>& inv = pass == 0 ? v : tmp;
        std::vector<T>& outv = pass == 0 ? tmp : v;
        std::fill(cnt.begin(), cnt.end(), 0);
        auto dig = [&](const T& t) { return pass == 0 ? (int)(t.k & 0xFFFFFFFFu) : (int)(t.k >> 32); };
        for (const T& t : inv) cnt[dig(t) + 1]++;
        for (int u = 0; u < n_free; ++u) cnt[u + 1] += cnt[u];
        for (const T& t : inv) outv[cnt[dig(t)]++] = t;
    }
}

// embedded mode: the observation lists of a plan's entries
static void nd_prep_ske(NdPrep& P, const NdPlan& PL) {
    NdStruct& T = *P.st;
    const int n_free = P.n_free;
    T.ske_ptr.assign(PL.ent.size() + 1, 0);
    T.ske_pt.clear(); T.ske_ia.clear(); T.ske_ib.clear();
    const size_t guess = 4 * P.nl_ix.size() + P.skt.size();
    T.ske_pt.reserve(guess); T.ske_ia.reserve(guess); T.ske_ib.reserve(guess);
    auto push = [&](int ia, int ib) { T.ske_pt.push_back(ia / SK_MAX); T.ske_ia.push_back(ia); T.ske_ib.push_back(ib); };
    for (size_t q = 0; q < PL.ent.size(); ++q) {
        const uint32_t kind = PL.ent[q].src >> ND_KIND_SHIFT, idx = PL.ent[q].src & ND_SRC_MASK;
        auto node_list = [&](int u, bool squared) {
            if (u >= n_free) return;
            for (int t = P.nl_ptr[u]; t < P.nl_ptr[u + 1]; ++t) push(P.nl_ix[t], squared ? P.nl_ix[t] : -1);
        };
        if (kind == 0) node_list((int)idx, true);
        else if (kind == 2) node_list((int)idx, false);
        else if (T.pkind[idx] == 0) { for (int t = P.pair_sk0[idx]; t < P.pair_sk1[idx]; ++t) push(P.skt[t].ia, P.skt[t].ib); }
        else if (T.pkind[idx] == 1) node_list(T.pairs[2 * (size_t)idx + 1], false);
        T.ske_ptr[q + 1] = (int)T.ske_pt.size();
    }
}
// ... and this frame's coefficients for them
static void nd_prep_ske_values(NdPrep& P, const double* sk_om) {
    const NdStruct& T = *P.st;
    const size_t n = T.ske_pt.size();
    P.ske_cf.resize(n);
    for (size_t t = 0; t < n; ++t) {
        const double a = sk_om[T.ske_ia[t]];
        P.ske_cf[t] = T.ske_ib[t] < 0 ? a : a * sk_om[T.ske_ib[t]];
    }
}

static void nd_prep_run_body(nrs_ctx* c, const NdIn& in, NdPrep& P);
// (runs on a helper thread of engine_create: nothing may escape it -- an exception there would end the process)
static void nd_prep_run(nrs_ctx* c, const NdIn& in, NdPrep& P) {
    try { nd_prep_run_body(c, in, P); }
    catch (const std::exception& ex) { P.wanted = false; P.plan_ok = false; P.hit = nullptr; P.err = ex.what(); }
    catch (...) { P.wanted = false; P.plan_ok = false; P.hit = nullptr; P.err = "unknown exception in the plan thread"; }
}
// halves of a dissection of this many nodes go to threads of their own (first two levels; NRS_ND_PLAN_PAR=0: never; NRS_HOST_THREADS=1 likewise)
static int nd_plan_par_min(const nrs_ctx* c) {
    if (const char* v = c->env("NRS_ND_PLAN_PAR")) return atoi(v);
    if (const char* v = c->env("NRS_HOST_THREADS")) if (atoi(v) <= 1) return 0;
    return std::thread::hardware_concurrency() >= 4 ? 700 : 0;
}
static void nd_prep_run_body(nrs_ctx* c, const NdIn& in, NdPrep& P) {
    P.wanted = false; P.plan_ok = false; P.hit = nullptr; P.st.reset();
    const bool tm = c->env("NRS_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    double t_ms[3] = {0, 0, 0};                                    // key + look-up, pairs, plan
    auto lap = [&](int k) { const auto now = std::chrono::steady_clock::now(); t_ms[k] = std::chrono::duration<double, std::milli>(now - t_prev).count(); t_prev = now; };
    P.node_of.assign(in.M, -1); P.node_vtx.clear();
    for (int v = 0; v < in.M; ++v)
        if (!(in.rflag[v] & RF_FIXED)) { P.node_of[v] = (int)P.node_vtx.size(); P.node_vtx.push_back(v); }
    const int n_free = P.n_free = (int)P.node_vtx.size();
    if (!nd_mode_allows(c, n_free)) return;
    for (int q = 0; q < in.n_dm; ++q)
        if (in.dm_idx[4 * (size_t)q] >= 0 || in.dm_idx[4 * (size_t)q + 1] >= 0) return;     // four-vertex dampers: a BA window, not this solver's problem
    P.pose_free = !in.pose_fixed;
    const int n_nodes = P.n_nodes = n_free + (P.pose_free ? 2 : 0);
    const std::vector<int>& node_of = P.node_of;
    // ---- the key: everything below (and the plan) depends on these arrays only -- and on the positions, which may be an earlier frame's
    {
        std::vector<uint8_t> bits(in.M);
        for (int v = 0; v < in.M; ++v) bits[v] = in.rflag[v] & (RF_FIXED | RF_OBS);
        const int hdr[8] = {n_free, P.pose_free ? 1 : 0, in.M, in.n_skin, nd_leaf_n(c), ND_SMAXN, in.n_sp, in.n_dm};
        std::vector<uint8_t>& key = P.key;
        key.clear();
        auto put = [&](const void* p, size_t bytes) { const uint8_t* b = static_cast<const uint8_t*>(p); key.insert(key.end(), b, b + bytes); };
        key.reserve(sizeof(hdr) + bits.size() + 8 * (size_t)in.n_sp + 16 * (size_t)in.n_dm + (size_t)SK_MAX * in.n_skin * 4);
        put(hdr, sizeof(hdr)); put(bits.data(), bits.size());
        put(in.sp_ij, 8 * (size_t)in.n_sp); put(in.dm_idx, 16 * (size_t)in.n_dm);
        if (in.n_skin > 0) put(in.sk_vert, 4 * (size_t)SK_MAX * in.n_skin);
        P.hash = nd_hash(key.data(), key.size());
    }
    // the cache (read only here: nobody changes it while an engine is being set up).  A frame whose key an earlier one had takes
    // that one's plan and structure as they are; only the skinning weights are its own.
    const NdCache* nc = static_cast<const NdCache*>(c->nd_cache);
    if (nc && !c->env("NRS_ND_NO_CACHE"))
        for (NdSlot* sl : nc->slots)
            if (!sl->busy && sl->st && sl->hash == P.hash && sl->key == P.key) {
                P.hit = sl; P.st = sl->st;
                if (in.n_skin > 0) nd_prep_ske_values(P, in.sk_om);
                P.wanted = true;
                return;
            }
    lap(0);
    P.st = std::make_shared<NdStruct>();
    NdStruct& T = *P.st;
    // unique row-row couplings with the edges that contribute to them
    struct Key { uint64_t k; int id; };
    std::vector<Key> keys;
    keys.reserve((size_t)in.n_sp + in.n_dm);
    auto add = [&](int va, int vb, int id) {
        const int a = node_of[va], b = node_of[vb];
        if (a < 0 || b < 0 || a == b) return;
        keys.push_back(Key{((uint64_t)std::min(a, b) << 32) | (uint32_t)std::max(a, b), id});
    };
    // (a2 gives every regulariser as a spring AND a damper over the same two vertices, index for index, OPT:281-335: one key per edge
    // then stands for both -- half the records to sort; inside a pair the springs come first and the dampers after them, as they arrive)
    bool twin = in.n_sp == in.n_dm;
    for (int q = 0; twin && q < in.n_sp; ++q)
        twin = in.sp_ij[2 * (size_t)q] == in.dm_idx[4 * (size_t)q + 2] && in.sp_ij[2 * (size_t)q + 1] == in.dm_idx[4 * (size_t)q + 3];
    for (int q = 0; q < in.n_sp; ++q) add(in.sp_ij[2 * (size_t)q], in.sp_ij[2 * (size_t)q + 1], q << 1);
    if (!twin) for (int q = 0; q < in.n_dm; ++q) add(in.dm_idx[4 * (size_t)q + 2], in.dm_idx[4 * (size_t)q + 3], (q << 1) | 1);
    // embedded mode: the node pairs every skinned observation couples (all pairs of its <= 11 free nodes), with the products of
    // its weights, and per free node the observations that reach it; everything in observation order (fixed summation order)
    std::vector<NdSkT>& skt = P.skt;
    skt.clear();
    P.nl_ptr.assign(n_free + 1, 0); P.nl_ix.clear();
    if (in.n_skin > 0) {
        std::vector<int>& nl_ptr = P.nl_ptr;
        size_t n_pairs_sk = 0;
        for (int i = 0; i < in.n_skin; ++i) {
            int cnt = 0;
            for (int a = 0; a < SK_MAX; ++a) {
                const int va = in.sk_vert[(size_t)SK_MAX * i + a];
                if (va >= 0 && node_of[va] >= 0) { nl_ptr[node_of[va] + 1]++; ++cnt; }
            }
            n_pairs_sk += (size_t)cnt * (cnt - 1) / 2;
        }
        for (int u = 0; u < n_free; ++u) nl_ptr[u + 1] += nl_ptr[u];
        P.nl_ix.resize(nl_ptr[n_free]);
        std::vector<int> fill(nl_ptr.begin(), nl_ptr.end() - 1);
        std::vector<NdSkT> raw;
        raw.reserve(n_pairs_sk);
        for (int i = 0; i < in.n_skin; ++i)
            for (int a = 0; a < SK_MAX; ++a) {
                const int va = in.sk_vert[(size_t)SK_MAX * i + a];
                if (va < 0 || node_of[va] < 0) continue;
                const int na = node_of[va];
                P.nl_ix[fill[na]++] = SK_MAX * i + a;
                for (int b = a + 1; b < SK_MAX; ++b) {
                    const int vb = in.sk_vert[(size_t)SK_MAX * i + b];
                    if (vb < 0 || node_of[vb] < 0 || node_of[vb] == na) continue;
                    const int nb2 = node_of[vb];
                    raw.push_back(NdSkT{((uint64_t)std::min(na, nb2) << 32) | (uint32_t)std::max(na, nb2), SK_MAX * i + a, SK_MAX * i + b});
                }
            }
        nd_sort_by_pair(raw, n_free);                               // by (low node, high node), observation order inside
        skt.swap(raw);
    }
    nd_sort_by_pair(keys, n_free);                                 // (edge order inside a pair)
    // the union of the regularisers' couplings and the observations': a merge of the two sorted key sequences (the arrays are sized for
    // the most there can be and written by index -- seven vector appends a pair were a third of this phase -- then cut to size)
    const size_t pairs_max = keys.size() + skt.size() + (P.pose_free ? 2 * (size_t)n_free + 1 : 0);
    T.pairs.resize(2 * pairs_max); T.pkind.resize(pairs_max); T.eptr.resize(pairs_max + 1); T.eid.resize(keys.size() * (twin ? 2 : 1));
    P.pair_sk0.resize(pairs_max); P.pair_sk1.resize(pairs_max);
    size_t np = 0, ne = 0;
    T.eptr[0] = 0;
    for (size_t i = 0, st = 0; i < keys.size() || st < skt.size();) {
        const uint64_t kk = i < keys.size() && (st >= skt.size() || keys[i].k <= skt[st].k) ? keys[i].k : skt[st].k;
        const size_t i0 = i;
        for (; i < keys.size() && keys[i].k == kk; ++i) T.eid[ne++] = keys[i].id;
        if (twin) for (size_t j = i0; j < i; ++j) T.eid[ne++] = keys[j].id | 1;
        T.eptr[np + 1] = (int)ne;
        T.pairs[2 * np] = (int)(kk >> 32); T.pairs[2 * np + 1] = (int)(kk & 0xFFFFFFFFu);
        T.pkind[np] = 0;
        P.pair_sk0[np] = (int)st;
        while (st < skt.size() && skt[st].k == kk) ++st;
        P.pair_sk1[np] = (int)st;
        ++np;
    }
    const size_t n_coupl = np;                                     // (the pose's pairs below carry no edge list and no observation range)
    P.last.assign(n_nodes, 0);
    if (P.pose_free) {
        P.last[n_free] = P.last[n_free + 1] = 1;
        for (int a = 0; a < n_free; ++a)
            if (in.rflag[P.node_vtx[a]] & RF_OBS)
                for (int h = 0; h < 2; ++h) { T.pkind[np] = 1; T.pairs[2 * np] = n_free + h; T.pairs[2 * np + 1] = a; ++np; }
        T.pkind[np] = 2; T.pairs[2 * np] = n_free + 1; T.pairs[2 * np + 1] = n_free; ++np;
    }
    T.pairs.resize(2 * np); T.pkind.resize(np); T.eptr.resize(n_coupl + 1); T.eid.resize(ne); P.pair_sk0.resize(n_coupl); P.pair_sk1.resize(n_coupl);
    P.wanted = true;
    lap(1);
    std::vector<double> pos(3 * (size_t)n_nodes, 0.0);
    for (int a = 0; a < n_free; ++a)
        for (int k = 0; k < 3; ++k) pos[3 * (size_t)a + k] = in.vpos[3 * (size_t)P.node_vtx[a] + k];
    P.plan_ok = nd_build_plan(n_nodes, pos.data(), P.last.data(), (int)T.pkind.size(), T.pairs.data(), P.plan, &P.err, nd_leaf_n(c), ND_SMAXN, false, nd_plan_par_min(c), c->env("NRS_ND_NO_COVER") == nullptr);
    if (P.plan_ok && in.n_skin > 0) { nd_prep_ske(P, P.plan); nd_prep_ske_values(P, in.sk_om); }
    lap(2);
    if (tm) fprintf(stderr, "[nrs] direct solve set-up thread: key %.2f ms, pairs %.2f ms, plan %.2f ms\n", t_ms[0], t_ms[1], t_ms[2]);
}

// phase B: leaves nd->on = false if the problem does not qualify
static int nd_engine_finish(nrs_ctx* c, Engine* e, NdEngine* nd, NdPrep& P) {
    Dev& d = e->d;
    {
        const bool tm = c->env("NRS_TIMING") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        P.wait();
        if (tm) fprintf(stderr, "[nrs] direct solve: waited %.2f ms for the plan thread\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    const bool tmf = c->env("NRS_TIMING") != nullptr;
    auto tf_prev = std::chrono::steady_clock::now();
    auto lapf = [&](const char* what) {
        if (!tmf) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[nrs] direct solve finish: %-18s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tf_prev).count());
        tf_prev = now;
    };
    nd_slot_release(c, nd);                                        // (a rebuild after the fixed set changed: the old plan goes back to the cache)
    if (!P.wanted || !nd_wanted(c, d, P.n_free)) return NRS_OK;
    const NdStruct& T = *P.st;
    const int n_free = P.n_free, n_nodes = P.n_nodes, n_pairs = (int)T.pkind.size();
    NdCache* nc = static_cast<NdCache*>(c->nd_cache);
    if (!nc) { nc = new (std::nothrow) NdCache(); if (!nc) return c->fail(NRS_ERR_ALLOC, "out of host memory"); c->nd_cache = nc; }
    const bool use_cache = true;                                  // (NRS_ND_NO_CACHE=1 only stops plans from being REUSED -- nd_prep_run -- the slots' buffers are)
    NdSlot* sl = P.hit && !P.hit->busy ? P.hit : nullptr;
    const bool hit = sl != nullptr;
    if (!hit) {
        if (P.hit) return c->fail(NRS_ERR_STATE, "direct solve: the plan this engine was set up on was taken by another engine meanwhile (engines of one context are created one at a time)");
        if (!P.plan_ok) {
            if (c->env("NRS_TIMING")) fprintf(stderr, "[nrs] direct solve not used: %s\n", P.err.c_str());
            return NRS_OK;
        }
        ++nc->misses;
        // a slot for the new plan: a free cached one (the least recently used is overwritten) or, if every cached slot is held by a
        // live engine, one that lives as long as this engine
        if (use_cache) {
            // a new slot while there is room; then the least recently used one among those whose buffer already holds this plan (the
            // factor and assembly storage of a 4.5k-point frame is ~100 MB: a hipMalloc per frame costs milliseconds); else the largest
            if ((int)nc->slots.size() < ND_CACHE_SLOTS) {
                sl = new (std::nothrow) NdSlot();
                if (!sl) return c->fail(NRS_ERR_ALLOC, "out of host memory");
                sl->cached = true;
                nc->slots.push_back(sl);
            } else {
                const size_t need = 8 * (P.plan.L_doubles + P.plan.A_doubles);
                for (NdSlot* q : nc->slots)
                    if (!q->busy && q->ws.cap >= need + need / 16 && (!sl || q->used < sl->used)) sl = q;
                if (!sl)
                    for (NdSlot* q : nc->slots)
                        if (!q->busy && (!sl || q->ws.cap > sl->ws.cap)) sl = q;
            }
        }
        if (!sl) { sl = new (std::nothrow) NdSlot(); if (!sl) return c->fail(NRS_ERR_ALLOC, "out of host memory"); }
    } else {
        ++nc->hits;
        if (c->env("NRS_TIMING")) fprintf(stderr, "[nrs] direct solve: plan of an earlier frame reused (%llu reused, %llu built)\n", (unsigned long long)nc->hits, (unsigned long long)nc->misses);
    }
    struct SlotGuard {                                             // a slot whose set-up fails holds nothing valid
        nrs_ctx* c; NdSlot* sl; bool keep = false;
        ~SlotGuard() { if (keep) return; if (sl->cached) { sl->hash = 0; sl->key.clear(); sl->key.push_back(0xFF); sl->used = 0; } else nd_slot_free(c, sl); }
    } sguard{c, sl};
    if (!hit) {
        sl->hash = 0; sl->key.clear(); sl->key.push_back(0xFF);      // (matches no key while it is rebuilt)
        sl->S.buf = &sl->ws;
        sl->S.n_alt = e->n_spec;                                   // (solve sets for the speculative trials: engine_optimize uses min(e->n_spec, n_alt))
        sl->S.plan = std::move(P.plan);
        const int up = nd_upload(c, sl->S);
        if (up == NRS_ERR_INVALID) {                               // a front or a boundary beyond the LDS: like a plan that could not be built -- the PCG takes the problem
            if (c->env("NRS_TIMING")) fprintf(stderr, "[nrs] direct solve not used: %s\n", c->err);
            return NRS_OK;                                         // (sguard leaves the slot empty; nd->on stays false; the embedded mode reports it, engine_create)
        }
        if (up != NRS_OK) return up;
    }
    lapf("plan upload");
    // ---- value descriptors: the plan's nodes and pairs in terms of this engine's rows and incidence slots
    std::vector<int> nrow(n_nodes), node_out(n_nodes);
    for (int a = 0; a < n_free; ++a) { nrow[a] = e->vrow[P.node_vtx[a]]; node_out[a] = 3 * nrow[a]; }
    if (P.pose_free) { nrow[n_free] = -1; nrow[n_free + 1] = -2; node_out[n_free] = -1; node_out[n_free + 1] = -1 - 3; }
    std::vector<NdPairD> pd(n_pairs);
    std::vector<int> src(T.eid.size());                             // (one source per edge of a pair, written by index)
    size_t n_src = 0;
    for (int i = 0; i < n_pairs; ++i) {
        const int a = T.pairs[2 * (size_t)i], b = T.pairs[2 * (size_t)i + 1];
        if (T.pkind[i] == 0) {
            // (the factor of an edge sits in both endpoints' incidence slots with the same value when both are free: the first one is read)
            pd[i] = NdPairD{0, nrow[a], nrow[b], (int)n_src, 0};
            for (int t = T.eptr[i]; t < T.eptr[i + 1]; ++t) {
                const int id = T.eid[t] >> 1, kind = T.eid[t] & 1;
                const int slot = kind ? e->dm_pos[4 * (size_t)id + 2] : e->sp_pos[2 * (size_t)id];
                if (slot < 0) return NRS_OK;                       // (an incidence of another rank: not a single-frame engine)
                src[n_src++] = (slot << 1) | kind;
            }
            pd[i].nsrc = (int)n_src - pd[i].src0;
        } else if (T.pkind[i] == 1) pd[i] = NdPairD{1, a - n_free, nrow[b], 0, 0};
        else pd[i] = NdPairD{2, 0, 0, 0, 0};
    }
    // embedded mode: per plan entry the observations that add to it (built beside the plan, nd_prep_run)
    const std::vector<int>&ske_ptr = T.ske_ptr, &ske_pt = T.ske_pt;
    const std::vector<double>& ske_cf = P.ske_cf;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_nr = 0, o_no = o_nr + al(4 * (size_t)n_nodes), o_pd = o_no + al(4 * (size_t)n_nodes), o_src = o_pd + al(sizeof(NdPairD) * (size_t)n_pairs),
                 o_sp = o_src + al(4 * std::max<size_t>(1, src.size())), o_st = o_sp + al(4 * std::max<size_t>(1, ske_ptr.size())),
                 o_sc = o_st + al(4 * std::max<size_t>(1, ske_pt.size())), total = o_sc + al(8 * std::max<size_t>(1, ske_cf.size()));
    lapf("descriptors");
    NRS_TRY(c->ensure(sl->vb, total));
    char* vb = sl->vb.as<char>();
    sl->h_vals.assign(total, 0);                                   // one upload from a staging image the slot keeps (no synchronisation)
    auto put = [&](size_t o, const void* srcp, size_t bytes) { if (bytes) memcpy(sl->h_vals.data() + o, srcp, bytes); };
    put(o_nr, nrow.data(), 4 * (size_t)n_nodes);
    put(o_no, node_out.data(), 4 * (size_t)n_nodes);
    put(o_pd, pd.data(), sizeof(NdPairD) * (size_t)n_pairs);
    put(o_src, src.data(), 4 * src.size());
    sl->vals.ske_ptr = nullptr; sl->vals.ske_pt = nullptr; sl->vals.ske_coef = nullptr;
    if (d.sk_n > 0) {
        put(o_sp, ske_ptr.data(), 4 * ske_ptr.size());
        put(o_st, ske_pt.data(), 4 * ske_pt.size());
        put(o_sc, ske_cf.data(), 8 * ske_cf.size());
        sl->vals.ske_ptr = reinterpret_cast<const int*>(vb + o_sp); sl->vals.ske_pt = reinterpret_cast<const int*>(vb + o_st);
        sl->vals.ske_coef = reinterpret_cast<const double*>(vb + o_sc);
    }
    NRS_HIP(c, hipMemcpyAsync(vb, sl->h_vals.data(), total, hipMemcpyHostToDevice, c->stream));
    sl->vals.node_row = reinterpret_cast<const int*>(vb + o_nr);
    sl->vals.pair = reinterpret_cast<const NdPairD*>(vb + o_pd);
    sl->vals.src = reinterpret_cast<const int*>(vb + o_src);
    sl->vals.ent = sl->S.d_ent; sl->vals.ev = sl->S.d_ev; sl->vals.n_ent = (int)sl->S.plan.ent.size();
    sl->S.dev.node_out = reinterpret_cast<const int*>(vb + o_no);
    sl->n_free = n_free; sl->n_pairs = n_pairs;
    // the engine's own vectors: where the solved step goes, the status words
    sl->S.dev.out_rows = d.xv; sl->S.dev.out_pose = d.xp; sl->S.dev.flags = d.flags;
    // rows the solver never writes (fixed, padding) keep a zero step; so does a fixed pose
    NRS_HIP(c, hipMemsetAsync(d.xv, 0, sizeof(double) * 3 * (size_t)d.n_rows, c->stream));
    NRS_HIP(c, hipMemsetAsync(d.xp, 0, sizeof(double) * 6 * (size_t)d.K, c->stream));
    for (int j = 0; j < e->n_spec; ++j) {
        NRS_HIP(c, hipMemsetAsync(e->spec[j].xv, 0, sizeof(double) * 3 * (size_t)d.n_rows, c->stream));
        NRS_HIP(c, hipMemsetAsync(e->spec[j].xp, 0, sizeof(double) * 6 * (size_t)d.K, c->stream));
    }
    nd->sig.assign((size_t)d.M + 1, 0);
    for (int v = 0; v < d.M; ++v) nd->sig[v] = e->h_rflag[e->vrow[v]] & RF_FIXED;
    nd->sig[d.M] = e->h_pose_fixed[0];
    lapf("values upload");
    sl->busy = true; sl->used = ++nc->clock;
    nd->slot = sl; nd->on = true;
    if (!hit) { sl->key.swap(P.key); sl->hash = P.hash; sl->st = P.st; }
    sguard.keep = true;
    if (c->env("NRS_TIMING"))
        fprintf(stderr, "[nrs] direct solve: %d free rows, %d pairs, %d fronts on %d levels, %d workgroups, %.1f MFLOP per factorisation\n", sl->n_free,
                sl->n_pairs, sl->S.plan.n_fronts, sl->S.plan.n_levels, (int)sl->S.plan.wg.size() / 3, sl->S.plan.flops / 1e6);
    return NRS_OK;
}

// both phases at once, from the engine's own mirrors (a rebuild after the fixed set changed)
static int nd_engine_setup(nrs_ctx* c, Engine* e, NdEngine* nd) {
    Dev& d = e->d;
    std::vector<uint8_t> rf(d.M);
    for (int v = 0; v < d.M; ++v) rf[v] = e->h_rflag[e->vrow[v]];
    NdIn in;
    in.M = d.M; in.rflag = rf.data(); in.pose_fixed = e->h_pose_fixed[0] != 0;
    in.n_sp = (int)(e->sp_ij.size() / 2); in.sp_ij = e->sp_ij.data();
    in.n_dm = (int)(e->dm_idx.size() / 4); in.dm_idx = e->dm_idx.data();
    in.n_skin = d.sk_n; in.sk_vert = e->sk_vert.data(); in.sk_om = e->sk_om.data();
    in.vpos = nd->pos.data();
    NdPrep P;
    nd_prep_run(c, in, P);
    return nd_engine_finish(c, e, nd, P);
}

static void nd_engine_free(nrs_ctx* c, NdEngine* nd) { nd_slot_release(c, nd); delete nd; }

}  // namespace nrs
