// EXPERIMENT (tools/micro/sweep_blk_probe.hip only; not part of the library): the 64-pivot sweep of csrc/nrs_kft_sweep.hpp with ONE wave for the
// chains.  Bit-identical to kft_sweep64_blk and SLOWER (29.8 k against 27.5 k cycles; 25.7 k with the chain wave alone on its SIMD): a dependent
// v_mfma_f64_16x16x4 costs ~100 cycles, so the chain wave's own diagonal updates (24 / 16 / 8 dependent matrix instructions between two chains)
// take 2.5 - 4.5 k cycles a step -- as long as the chains they were meant to hide -- and the main waves' steps are bound by the same latency
// (stamps: -DKFT_EXP_STAMPS).  Interleaving a step's matrix instructions as four independent chains (k-step outermost) did not help either:
// 28.8 k cycles, 264 registers.
// (included inside namespace nrs, after nrs_kft_sweep.hpp)
// ---- the same sweep with ONE wave for the chains (workgroups of eight waves).  In kft_sweep64_blk every wave factorises every diagonal block
// for itself, and a step's chain (3 k cycles) and its matrix instructions (3.7 k) alternate: 27 k cycles.  Here wave 4 keeps the four diagonal
// tiles of its own, updates them itself (D_j -= Y_j^T Y_j from the step's row panel: the arithmetic wave j does on its copy, bit for bit) and so
// starts the next block's chain the moment this one is done: the chains run back to back (12 k cycles) while waves 0 - 3 do the panels' matrix
// work behind them.  Hand-over inside the workgroup through LDS flags (release / acquire at workgroup scope, bounded waits): W_K is posted by
// the chain wave, the row panel R_K by wave K as soon as its rows are up to date.  One barrier at the start, none after.
// xb: KFT_CHAIN_XB doubles; every wave of the workgroup calls it with its wave index (0 .. 7); returns the bad-pivot flag to every wave
// that reaches the caller's next barrier (it is left in LDS: read it AFTER that barrier through kft_chain_bad).
constexpr int KFT_CHAIN_R = 0, KFT_CHAIN_W = 4 * 16 * KFT_RS, KFT_CHAIN_D = KFT_CHAIN_W + 4 * 16 * KFT_WS, KFT_CHAIN_F = KFT_CHAIN_D + 4 * 16 * KFT_WS;
constexpr int KFT_CHAIN_XB = KFT_CHAIN_F + 8;                      // (flags: 4 x W posted, 4 x R posted, bad, as ints in the last 8 doubles)
__device__ __forceinline__ void kft_flag_wait(int* f, int* badw) {
    int spins = 0;
#ifdef KFT_EXP_NOWAIT                                               // (probe: no hand-over waits at all -- wrong results, the main waves' own time)
    return;
#endif
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
#ifndef KFT_EXP_TIGHTPOLL
        __builtin_amdgcn_s_sleep(2);
#endif
        if (++spins > (1 << 22)) { __hip_atomic_store(badw, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }   // (cannot happen: a fault, not a matrix property)
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void kft_flag_post(int* f, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int K, int J>
__device__ __forceinline__ void kft_chain_diag_upd(KftTiles& dd, const double* R, int lc, int lk, const double (&ws)[4]) {   // D_j -= Y_j^T Y_j, j > K
    if constexpr (J < 4) {
        double b[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) b[s] = R[(lk + 4 * s) * KFT_RS + 16 * J + lc];
        nd_v4d y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) y = __builtin_amdgcn_mfma_f64_16x16x4f64(ws[s], b[s], y, 0, 0, 0);
        const nd_v4d ny = -y;
#pragma unroll
        for (int s = 0; s < 4; ++s) dd.at<J>() = __builtin_amdgcn_mfma_f64_16x16x4f64(ny[s], y[s], dd.at<J>(), 0, 0, 0);
        kft_chain_diag_upd<K, J + 1>(dd, R, lc, lk, ws);
    }
}
template <int K>
__device__ __forceinline__ void kft_chain_wave(KftTiles& dd, double* xb, int* flg, int lane, int& bad) {
    if constexpr (K < 4) {
        const int lc = lane & 15, lk = lane >> 4;
        double* Dk = xb + KFT_CHAIN_D + K * (16 * KFT_WS);
        double* Wk = xb + KFT_CHAIN_W + K * (16 * KFT_WS);
        if constexpr (K > 0) {                                     // the block as rows: through the wave's own area
#pragma unroll
            for (int g = 0; g < 4; ++g) Dk[(lk + 4 * g) * KFT_WS + lc] = dd.at<K>()[g];
            __builtin_amdgcn_wave_barrier();
        }
        double a[16], wv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = Dk[lc * KFT_WS + q];
        double mine = 1.0;
#ifdef KFT_EXP_STAMPS
        if (lane == 0) reinterpret_cast<long long*>(xb + KFT_CHAIN_XB)[2 * K] = clock64();
#endif
        kft_ldl16<0>(a, wv, mine, lc, bad);
#ifdef KFT_EXP_STAMPS
        if (lane == 0) reinterpret_cast<long long*>(xb + KFT_CHAIN_XB)[2 * K + 1] = clock64();
#endif
        const double rs = kft_rsqrt(mine);
#pragma unroll
        for (int q = 0; q < 16; ++q) wv[q] *= rs;
        if (lk == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) Wk[lc * KFT_WS + q] = wv[q];
        }
        kft_flag_post(flg + K, lane);
        if constexpr (K < 3) {
            double ws[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) ws[s] = Wk[lc * KFT_WS + 4 * s + lk];
            if constexpr (K > 0) kft_flag_wait(flg + 4 + K, flg + 8);
            kft_chain_diag_upd<K, K + 1>(dd, xb + KFT_CHAIN_R + K * (16 * KFT_RS), lc, lk, ws);
        }
        kft_chain_wave<K + 1>(dd, xb, flg, lane, bad);
    }
}
template <int K>
__device__ __forceinline__ void kft_chain_main(KftTiles& c, double* xb, int* flg, int lane, int w) {
    if constexpr (K < 4) {
        const int lc = lane & 15, lk = lane >> 4;
        const double* R = xb + KFT_CHAIN_R + K * (16 * KFT_RS);
        const double* Wk = xb + KFT_CHAIN_W + K * (16 * KFT_WS);
        kft_flag_wait(flg + K, flg + 8);
        if (K > 0 && w != K) kft_flag_wait(flg + 4 + K, flg + 8);
        double ws[4], wt[4], rw[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) { ws[s] = Wk[lc * KFT_WS + 4 * s + lk]; wt[s] = Wk[(4 * s + lk) * KFT_WS + lc]; rw[s] = R[(lk + 4 * s) * KFT_RS + 16 * w + lc]; }
        const nd_v4d zero = {0.0, 0.0, 0.0, 0.0};
        nd_v4d nyw = zero;
        if (w == K) {
            nd_v4d z = zero;
#pragma unroll
            for (int s = 0; s < 4; ++s) z = __builtin_amdgcn_mfma_f64_16x16x4f64(wt[s], wt[s], z, 0, 0, 0);
            c.at<K>() = -z;
        } else {
            nd_v4d yw = zero, ck = zero;
#pragma unroll
            for (int s = 0; s < 4; ++s) yw = __builtin_amdgcn_mfma_f64_16x16x4f64(ws[s], rw[s], yw, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 4; ++s) ck = __builtin_amdgcn_mfma_f64_16x16x4f64(yw[s], wt[s], ck, 0, 0, 0);
            nyw = -yw;
            c.at<K>() = ck;
        }
        kft_blk_tiles<K, 0>(c, R, lc, lk, ws, wt, nyw, w == K);
        if constexpr (K < 3) {
            if (w == K + 1) {                                      // the next pivot rows are up to date: their panel
                kft_blk_panel_out<K + 1, 0>(c, xb + KFT_CHAIN_R + (K + 1) * (16 * KFT_RS), lc, lk);
                kft_flag_post(flg + 4 + K + 1, lane);
            }
        }
        kft_chain_main<K + 1>(c, xb, flg, lane, w);
    }
}
__device__ __forceinline__ void kft_sweep64_chain(KftTiles& c, double* xb, int lane, int w) {   // w: 0 .. 7 (every wave of the workgroup)
    int* flg = reinterpret_cast<int*>(xb + KFT_CHAIN_F);
    const int lc = lane & 15, lk = lane >> 4;
    if (w == 7 && lane < 16) flg[lane] = 0;
    if (w < 4) {                                                   // a wave's diagonal tile for the chain wave; wave 0's rows: the first panel
        double* Dw = xb + KFT_CHAIN_D + w * (16 * KFT_WS);
        const nd_v4d dv = w == 0 ? c.t0 : w == 1 ? c.t1 : w == 2 ? c.t2 : c.t3;
#pragma unroll
        for (int g = 0; g < 4; ++g) Dw[(lk + 4 * g) * KFT_WS + lc] = dv[g];
        if (w == 0) kft_blk_panel_out<0, 0>(c, xb + KFT_CHAIN_R, lc, lk);
    }
    __syncthreads();
    if (w < 4) kft_chain_main<0>(c, xb, flg, lane, w);
    else if (w == 4) {
        KftTiles dd;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            dd.t0[g] = 0.0;                                        // (block 0 is read as rows from the waves' copy; the others as accumulator tiles)
            dd.t1[g] = xb[KFT_CHAIN_D + 1 * (16 * KFT_WS) + (lk + 4 * g) * KFT_WS + lc];
            dd.t2[g] = xb[KFT_CHAIN_D + 2 * (16 * KFT_WS) + (lk + 4 * g) * KFT_WS + lc];
            dd.t3[g] = xb[KFT_CHAIN_D + 3 * (16 * KFT_WS) + (lk + 4 * g) * KFT_WS + lc];
        }
        int bad = 0;
        kft_chain_wave<0>(dd, xb, flg, lane, bad);
        if (bad && lane == 0) __hip_atomic_store(flg + 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
__device__ __forceinline__ int kft_chain_bad(const double* xb) { return reinterpret_cast<const int*>(xb + KFT_CHAIN_F)[8]; }   // (after the barrier behind kft_sweep64_chain)
