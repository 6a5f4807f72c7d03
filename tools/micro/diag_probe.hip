// Cycle counts of the 16 x 16 diagonal-block Cholesky of the direct solver (one wave, row per lane, DPP row broadcasts) in isolation:
// variants of the column loop.  hipcc --offload-arch=gfx950 -O3 tools/micro/diag_probe.hip -o /tmp/diag_probe && /tmp/diag_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__device__ inline double frsq(double a) { const double y = __builtin_amdgcn_rsq(a); const double e = fma(-a * y, y, 1.0); return fma(y * e, fma(e, 0.375, 0.5), y); }
template <int K> __device__ inline double bc(double v) {
    return __builtin_amdgcn_update_dpp(v, v, 0x150 + K, 0xf, 0xf, false);
}
template <int K> __device__ inline void fmac_bc(double& a, double nl, double l) {      // a += (lane K's nl of this 16-lane row) * l
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(nl), "v"(l), "n"(K));
}
template <int J, int K> __device__ inline void upd(double (&a)[16], double nl, double l) { if constexpr (K < 16) { fmac_bc<K>(a[K], nl, l); upd<J, K + 1>(a, nl, l); } }
template <int V, int J> __device__ inline void cols(double (&a)[16], double (&rr)[16], int i, int& bad) {
    if constexpr (J < 16) {
        double ajj = bc<J>(a[J]);
        if (V == 0) { if (!(ajj > 0.0)) { bad = 1; ajj = 1.0; } }
        else { const bool ok = ajj > 0.0; bad |= !ok; ajj = ok ? ajj : 1.0; }
        const double r = frsq(ajj);
        const double l = (i == J) ? ajj * r : a[J] * r;
        a[J] = l; rr[J] = r;
        { double nl = -l; asm volatile("s_nop 1" : "+v"(nl)); upd<J, J + 1>(a, nl, l); }
        cols<V, J + 1>(a, rr, i, bad);
    }
}

// variant 2 (round 5): column J's pivot chain (broadcast, rsq + Newton step, scaling) INTERLEAVED with the pending updates of column
// J - 1 -- the wave issues in order, so in the recursive form above a column's ~15 independent FMAs and the next column's ~10 dependent
// chain operations run one after the other; here the order is pinned by scheduling barriers
#define SB __builtin_amdgcn_sched_barrier(0)
template <int J, int K> __device__ inline void pend(double (&a)[16], double pnl, double pl) {
    if constexpr (J > 0 && K < 16) fmac_bc<K>(a[K], pnl, pl);
}
template <int J> __device__ inline void colp(double (&a)[16], double (&rr)[16], int& bad, double pl, double pnl) {
    if constexpr (J < 16) {
        pend<J, J>(a, pnl, pl);                                    // a[J] is final
        SB;
        pend<J, J + 1>(a, pnl, pl); pend<J, J + 2>(a, pnl, pl);
        SB;
        double ajj = bc<J>(a[J]);
        SB;
        pend<J, J + 3>(a, pnl, pl);
        SB;
        const bool ok = ajj > 0.0; bad |= !ok; ajj = ok ? ajj : 1.0;
        SB;
        pend<J, J + 4>(a, pnl, pl);
        SB;
        const double y = __builtin_amdgcn_rsq(ajj);
        SB;
        pend<J, J + 5>(a, pnl, pl); pend<J, J + 6>(a, pnl, pl); pend<J, J + 7>(a, pnl, pl);
        SB;
        const double t = -ajj * y;
        SB;
        pend<J, J + 8>(a, pnl, pl); pend<J, J + 9>(a, pnl, pl);
        SB;
        const double e = fma(t, y, 1.0);
        SB;
        pend<J, J + 10>(a, pnl, pl); pend<J, J + 11>(a, pnl, pl);
        SB;
        const double ye = y * e, f = fma(e, 0.375, 0.5);
        SB;
        pend<J, J + 12>(a, pnl, pl); pend<J, J + 13>(a, pnl, pl);
        SB;
        const double r = fma(ye, f, y);
        SB;
        pend<J, J + 14>(a, pnl, pl); pend<J, J + 15>(a, pnl, pl);
        SB;
        const double l = a[J] * r;
        double nl = -l;
        asm volatile("s_nop 1" : "+v"(nl));
        a[J] = l; rr[J] = r;
        SB;
        colp<J + 1>(a, rr, bad, l, nl);
    }
}

// variant 3 (round 5): a shorter pivot chain -- the negation rides on the DPP operand (src0 neg modifier: no xor / move per column) and a
// bad pivot is replaced by changing its high word only (one select instead of two: any value in [1, 2) will do)
template <int K> __device__ inline void fmacn_bc(double& a, double lsrc, double l) {   // a -= (lane K's lsrc of this 16-lane row) * l
    asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(lsrc), "v"(l), "n"(K));
}
template <int J, int K> __device__ inline void updn(double (&a)[16], double l) { if constexpr (K < 16) { fmacn_bc<K>(a[K], l, l); updn<J, K + 1>(a, l); } }
template <int J> __device__ inline void cols3(double (&a)[16], double (&rr)[16], int& bad) {
    if constexpr (J < 16) {
        double ajj = bc<J>(a[J]);
        const bool ok = ajj > 0.0; bad |= !ok;
        ajj = __hiloint2double(ok ? __double2hiint(ajj) : 0x3FF00000, __double2loint(ajj));
        const double r = frsq(ajj);
        double l = a[J] * r;
        asm volatile("s_nop 1" : "+v"(l));
        a[J] = l; rr[J] = r;
        updn<J, J + 1>(a, l);
        cols3<J + 1>(a, rr, bad);
    }
}

// variant 4: 2 + 3 (the short chain, pinned between the pending updates of the column before)
template <int J, int K> __device__ inline void pendn(double (&a)[16], double pl) { if constexpr (J > 0 && K < 16) fmacn_bc<K>(a[K], pl, pl); }
template <int J> __device__ inline void colq(double (&a)[16], double (&rr)[16], int& bad, double pl) {
    if constexpr (J < 16) {
        pendn<J, J>(a, pl);
        SB;
        pendn<J, J + 1>(a, pl); pendn<J, J + 2>(a, pl);
        SB;
        double ajj = bc<J>(a[J]);
        SB;
        pendn<J, J + 3>(a, pl); pendn<J, J + 4>(a, pl);
        SB;
        const bool ok = ajj > 0.0; bad |= !ok;
        ajj = __hiloint2double(ok ? __double2hiint(ajj) : 0x3FF00000, __double2loint(ajj));
        SB;
        pendn<J, J + 5>(a, pl);
        SB;
        const double y = __builtin_amdgcn_rsq(ajj);
        SB;
        pendn<J, J + 6>(a, pl); pendn<J, J + 7>(a, pl); pendn<J, J + 8>(a, pl);
        SB;
        const double t = -ajj * y;
        SB;
        pendn<J, J + 9>(a, pl); pendn<J, J + 10>(a, pl);
        SB;
        const double e = fma(t, y, 1.0);
        SB;
        pendn<J, J + 11>(a, pl); pendn<J, J + 12>(a, pl);
        SB;
        const double ye = y * e, f = fma(e, 0.375, 0.5);
        SB;
        pendn<J, J + 13>(a, pl); pendn<J, J + 14>(a, pl);
        SB;
        const double r = fma(ye, f, y);
        SB;
        pendn<J, J + 15>(a, pl);
        SB;
        double l = a[J] * r;
        asm volatile("s_nop 1" : "+v"(l));
        a[J] = l; rr[J] = r;
        SB;
        colq<J + 1>(a, rr, bad, l);
    }
}
template <int V> __global__ void k(double* out, const double* in, long long* t, int reps) {
    __shared__ double W[16 * 17], dinv[16];
    const int lane = threadIdx.x & 63, i = lane & 15;
    for (int j = threadIdx.x; j < 256; j += blockDim.x) W[(j >> 4) * 17 + (j & 15)] = in[j];
    __syncthreads();
    long long tot = 0;
    int bad = 0;
    double a[16], rr[16];
    for (int rep = 0; rep < reps; ++rep) {
        const long long c0 = clock64();
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = W[i * 17 + j];
        if constexpr (V == 4) colq<0>(a, rr, bad, 0.0); else if constexpr (V == 3) cols3<0>(a, rr, bad); else if constexpr (V == 2) colp<0>(a, rr, bad, 0.0, 0.0); else cols<V, 0>(a, rr, i, bad);
        if (lane < 16) {
#pragma unroll
            for (int j = 0; j < 16; ++j) if (j <= i) out[i * 16 + j] = a[j];
        }
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) dinv[j] = rr[j];
        }
        tot += clock64() - c0;
        __syncthreads();
    }
    if (threadIdx.x == 0) { t[0] = tot; t[1] = bad; out[300] = dinv[3]; }
}
int main() {
    double h[256], *in, *out; long long* t;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) h[i * 16 + j] = (i == j ? 20.0 : 0.0) + 1.0 / (1 + abs(i - j));
    hipMalloc(&in, 2048); hipMalloc(&out, 4096); hipMalloc(&t, 64);
    hipMemcpy(in, h, 2048, hipMemcpyHostToDevice);
    for (int v = 0; v < 5; ++v) for (int rep = 0; rep < 2; ++rep) {
        if (v == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, out, in, t, 100); else if (v == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, out, in, t, 100);
        else if (v == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, out, in, t, 100);
        else if (v == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, out, in, t, 100);
        else hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, out, in, t, 100);
        long long ht[2]; double ho[256];
        hipMemcpy(ht, t, 16, hipMemcpyDeviceToHost); hipMemcpy(ho, out, 2048, hipMemcpyDeviceToHost);
        double L[16][16] = {}, err = 0;                                // host check
        for (int j = 0; j < 16; ++j) { double d = h[j * 16 + j]; for (int q = 0; q < j; ++q) d -= L[j][q] * L[j][q]; L[j][j] = sqrt(d);
            for (int i2 = j + 1; i2 < 16; ++i2) { double s = h[i2 * 16 + j]; for (int q = 0; q < j; ++q) s -= L[i2][q] * L[j][q]; L[i2][j] = s / L[j][j]; } }
        for (int i2 = 0; i2 < 16; ++i2) for (int j = 0; j <= i2; ++j) err = fmax(err, fabs(L[i2][j] - ho[i2 * 16 + j]));
        printf("variant %d: %.0f cycles per 16x16 block (%.2f us at 2.4 GHz), max err %.2e, bad %lld\n", v, ht[0] / 100.0, ht[0] / 100.0 / 2400.0, err, ht[1]);
    }
    return 0;
}
