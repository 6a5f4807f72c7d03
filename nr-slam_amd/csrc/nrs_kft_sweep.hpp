// The 64 x 64 pivot block's sweep of the keyframe-block factorisation (nrs_engine_kft.hpp, k_kft_step) in 16-pivot steps.  Kept in a file of its
// own so that tools/micro/sweep_blk_probe.hip can run it alone against a host inverse.  Needs nd_v4d, nd_rowbcast, nd_fmacn_bcast
// (nrs_engine_nd.hpp) and KFT_B.
#pragma once
// (included inside namespace nrs)

// The sweep of the 64 x 64 pivot block in FOUR 16-pivot steps, the block in the matrix cores' accumulator layout (c[n][g] = entry
// (16 w + (lane >> 4) + 4 g, 16 n + (lane & 15)) of wave w: a wave owns sixteen rows).  Step K: wave K leaves its rows -- the pivot row panel
// R = A[K, :] -- in LDS (one barrier per step, panels double-buffered); EVERY wave factorises the 16 x 16 diagonal block D = R[:, K] for itself,
// one row per lane on DPP row broadcasts: D = L Delta L^T by rows, the same row operations applied to the identity, so that W = Delta^-1/2 L^-1
// (D^-1 = W^T W) is what a lane ends with -- per pivot a reciprocal, one multiply and fifteen v_fmac_f64_dpp on the dependent chain, where the
// 4-pivot register form (kft_sweep64) spends ~190 vector operations per four pivots plus an LDS round trip.  Then, on v_mfma_f64_16x16x4,
//     Y = W R  (the Cholesky form of the panel: every wave computes the tiles it needs),   A[w, n] -= Y_w^T Y_n  (n != K),
//     A[w, K] = Y_w^T W,   and for wave K itself   A[K, n] = W^T Y_n,   A[K, K] = -W^T W
// (Y_w = Y[:, w]: the matrix is symmetric, the wave's column tile is read from the row panel; an accumulator tile IS the A operand of its
// transpose, so only W^T goes through LDS, inside the wave).  The products are formed through W, not through an explicit D^-1: with D^-1
// the result loses cond(D) where this form and the pivot-by-pivot sweep lose its square root (tools/micro/sweep_blk_probe.hip; a window's blocks at
// lambda = 1e-5 max diag: 1e-7 against 5e-13 in M^-1 (H + lambda I) x = x).  Leaves -A^-1 in c like kft_sweep64; same result up to the
// association of the updates.  xb: KFT_SWEEP_XB doubles.
constexpr int KFT_RS = KFT_B + 2;    // row stride of the pivot row panel in LDS
constexpr int KFT_WS = 17;           // row stride of a wave's copy of W
constexpr int KFT_SWEEP_XB = 2 * 16 * KFT_RS + 4 * 16 * KFT_WS;
__device__ inline double kft_rsqrt(double a) {                     // a^-1/2, a > 0 normal: v_rsq_f64 + one third-order step
    const double y = __builtin_amdgcn_rsq(a);
    const double e = fma(-a * y, y, 1.0);
    return fma(y * e, fma(e, 0.375, 0.5), y);
}
template <int J, int K0, int K1>
__device__ inline void kft_ldl_upd(double (&a)[16], double m) {    // a[k] -= (lane J's a[k]) * m for k in [K0, K1)
    if constexpr (K0 < K1) {
        nd_fmacn_bcast<J>(a[K0], a[K0], m);
        kft_ldl_upd<J, K0 + 1, K1>(a, m);
    }
}
// lane i (of every 16-lane row) holds row i of D in a; leaves row i of L^-1 (unit lower triangular; column J is created by step J) in wv and
// the lane's own pivot in mine
template <int J>
__device__ inline void kft_ldl16(double (&a)[16], double (&wv)[16], double& mine, int i, int& bad) {
    if constexpr (J < 16) {
        asm volatile("" : "+v"(i));                                // (opaque: the row masks are formed per pivot instead of 32 compare results held -- and spilled -- across the steps)
        double d = nd_rowbcast<J>(a[J]);
        const bool ok = (d > 0.0) & (d < 1e300);
        bad = ok ? bad : 1;                                        // (a vector select per pivot: as `bad |= !ok` the sixty-four compare results are kept in scalar registers, and spilled, to be combined at the end)
        asm volatile("" : "+v"(bad));
        d = __hiloint2double(ok ? __double2hiint(d) : 0x3FF00000, __double2loint(d));   // (a bad pivot: any value in [1, 2) will do -- one select)
        double inv = __builtin_amdgcn_rcp(d);
        inv = fma(fma(-d, inv, 1.0), inv, inv);
        const int below = (J - i) >> 31, own = ((J ^ i) - 1) >> 31;                      // all ones in the rows below the pivot / in the pivot's row (masks: no
        mine = __hiloint2double((__double2hiint(d) & own) | (__double2hiint(mine) & ~own), (__double2loint(d) & own) | (__double2loint(mine) & ~own));   // compare results kept in scalar registers across the unrolled steps)
        const double m = a[J] * __hiloint2double(__double2hiint(inv) & below, __double2loint(inv) & below);   // (rows above the pivot are finished: 0)
        kft_ldl_upd<J, J + 1, 16>(a, m);
        kft_ldl_upd<J, 0, J>(wv, m);
        wv[J] = __hiloint2double(((__double2hiint(m) ^ 0x80000000) & ~own) | (0x3FF00000 & own), __double2loint(m) & ~own);   // column J of L^-1 so far: 1 in the pivot's row, -m below, 0 above
        asm volatile("s_nop 1" : "+v"(wv[J]));                     // (read through DPP by the next steps)
        kft_ldl16<J + 1>(a, wv, mine, i, bad);
    }
}
// the wave's sixteen rows as four accumulator tiles by NAME (a tile array indexed inside the unrolled loops stayed in scratch memory: six
// 16-byte scratch loads and stores per step)
struct KftTiles {
    nd_v4d t0, t1, t2, t3;
    template <int N> __device__ __forceinline__ nd_v4d& at() {
        if constexpr (N == 0) return t0; else if constexpr (N == 1) return t1; else if constexpr (N == 2) return t2; else return t3;
    }
};
template <int K, int N>
__device__ __forceinline__ void kft_blk_panel_out(KftTiles& c, double* R, int lc, int lk) {   // wave K: its rows to the panel
    if constexpr (N < 4) {
#pragma unroll
        for (int g = 0; g < 4; ++g) R[(lk + 4 * g) * KFT_RS + 16 * N + lc] = c.at<N>()[g];
        kft_blk_panel_out<K, N + 1>(c, R, lc, lk);
    }
}
template <int K, int N>
__device__ __forceinline__ void kft_blk_tiles(KftTiles& c, const double* R, int lc, int lk, const double (&ws)[4], const double (&wt)[4], const nd_v4d& nyw, bool pivot_wave) {
    if constexpr (N < 4) {
        if constexpr (N != K) {
            double b[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) b[s] = R[(lk + 4 * s) * KFT_RS + 16 * N + lc];
            nd_v4d y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) y = __builtin_amdgcn_mfma_f64_16x16x4f64(ws[s], b[s], y, 0, 0, 0);          // Y_N = W R_N
            if (pivot_wave) {
                nd_v4d z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 4; ++s) z = __builtin_amdgcn_mfma_f64_16x16x4f64(wt[s], y[s], z, 0, 0, 0);      // A[K, N] = W^T Y_N
                c.at<N>() = z;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) c.at<N>() = __builtin_amdgcn_mfma_f64_16x16x4f64(nyw[s], y[s], c.at<N>(), 0, 0, 0);   // A[w, N] -= Y_w^T Y_N
            }
        }
        kft_blk_tiles<K, N + 1>(c, R, lc, lk, ws, wt, nyw, pivot_wave);
    }
}
template <int K>
__device__ __forceinline__ void kft_sweep_blk_steps(KftTiles& c, double* xb, int lane, int w, int& bad) {
    if constexpr (K < 4) {
        double* R = xb + (K & 1) * (16 * KFT_RS);
        double* Wl = xb + 2 * 16 * KFT_RS + w * (16 * KFT_WS);
        const int lc = lane & 15, lk = lane >> 4;
        if (w == K) kft_blk_panel_out<K, 0>(c, R, lc, lk);
        __syncthreads();
        double a[16], wv[16], rw[4];
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = R[lc * KFT_RS + 16 * K + q];
#pragma unroll
        for (int s = 0; s < 4; ++s) rw[s] = R[(lk + 4 * s) * KFT_RS + 16 * w + lc];
        double mine = 1.0;
#ifndef KFT_EXP_NOLDL                                              // (tools/micro/sweep_blk_probe.hip: the sweep without its chains, timing only)
        kft_ldl16<0>(a, wv, mine, lc, bad);
#else
#pragma unroll
        for (int q = 0; q < 16; ++q) wv[q] = a[q];
#endif
        const double rs = kft_rsqrt(mine);
#pragma unroll
        for (int q = 0; q < 16; ++q) wv[q] *= rs;                  // row lc of W
        if (lk == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) Wl[lc * KFT_WS + q] = wv[q];
        }
        double ws[4], wt[4];                                       // W[lc][4 s + lk], W[4 s + lk][lc]: both from the wave's copy (a select chain over
        __builtin_amdgcn_wave_barrier();                           // lk compiles to divergent branches); LDS operations of a wave complete in order
#pragma unroll
        for (int s = 0; s < 4; ++s) { ws[s] = Wl[lc * KFT_WS + 4 * s + lk]; wt[s] = Wl[(4 * s + lk) * KFT_WS + lc]; }
        const nd_v4d zero = {0.0, 0.0, 0.0, 0.0};
        nd_v4d nyw = zero;
        if (w == K) {
            nd_v4d z = zero;
#pragma unroll
            for (int s = 0; s < 4; ++s) z = __builtin_amdgcn_mfma_f64_16x16x4f64(wt[s], wt[s], z, 0, 0, 0);
            c.at<K>() = -z;                                        // A[K, K] = -W^T W
        } else {
            nd_v4d yw = zero, ck = zero;
#pragma unroll
            for (int s = 0; s < 4; ++s) yw = __builtin_amdgcn_mfma_f64_16x16x4f64(ws[s], rw[s], yw, 0, 0, 0);       // Y_w
#pragma unroll
            for (int s = 0; s < 4; ++s) ck = __builtin_amdgcn_mfma_f64_16x16x4f64(yw[s], wt[s], ck, 0, 0, 0);       // A[w, K] = Y_w^T W
            nyw = -yw;
            c.at<K>() = ck;
        }
        kft_blk_tiles<K, 0>(c, R, lc, lk, ws, wt, nyw, w == K);
        kft_sweep_blk_steps<K + 1>(c, xb, lane, w, bad);
    }
}
__device__ __forceinline__ bool kft_sweep64_blk(KftTiles& c, double* xb, int lane, int w) {
    int bad = 0;
    kft_sweep_blk_steps<0>(c, xb, lane, w, bad);
    return bad != 0;
}
