// Latencies that bound single-wave serial code on gfx950 (direct solver's diagonal blocks / back substitution): core clock during a short
// kernel, dependent fp64 FMA chain, independent fp64 FMAs, v_readlane -> FMA, LDS broadcast read -> FMA, f64 MFMA chain / independent.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lat_probe.hip -o /tmp/lat_probe && /tmp/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__device__ inline double rl(double v, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__global__ void k(double* out, long long* t, int n) {
    __shared__ double lds[256];
    const int lane = threadIdx.x & 63;
    lds[threadIdx.x] = 1.0 + 1e-9 * threadIdx.x;
    __syncthreads();
    double x = 1.0 + lane * 1e-9, y = 0.999999;
    long long w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) x = fma(x, y, 1e-12);               // dependent chain
    }
    long long c1 = clock64();
    double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 2; ++u) { a0 = fma(a0, y, 1e-12); a1 = fma(a1, y, 1e-12); a2 = fma(a2, y, 1e-12); a3 = fma(a3, y, 1e-12); a4 = fma(a4, y, 1e-12); a5 = fma(a5, y, 1e-12); a6 = fma(a6, y, 1e-12); a7 = fma(a7, y, 1e-12); }
    }
    long long c2 = clock64();
    x = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) x = fma(x, rl(x, u), 1e-12);        // readlane of the fresh value -> FMA (the diagonal block's chain)
    }
    long long c3 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int idx = (__double2loint(x) & 0) + u + (i & 7); x = fma(x, lds[idx], 1e-12); }   // LDS broadcast read depending on x -> FMA
    }
    long long c4 = clock64();
    v4d acc = {x, x, x, x};
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, acc, 0, 0, 0);   // dependent MFMA chain
    }
    long long c5 = clock64();
    v4d b0 = acc, b1 = acc, b2 = acc, b3 = acc;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { b0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, b0, 0, 0, 0); b1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, b1, 0, 0, 0); b2 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, b2, 0, 0, 0); b3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, b3, 0, 0, 0); }
    }
    long long c6 = clock64();
    double r = x;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) r = __builtin_amdgcn_rsq(r + 1.5);   // dependent rsq + add
    }
    long long c7 = clock64();
    for (int i = 0; i < n; ++i) { __syncthreads(); }
    long long c8 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = w1 - w0; t[1] = c8 - c0; t[2] = c1 - c0; t[3] = c2 - c1; t[4] = c3 - c2; t[5] = c4 - c3; t[6] = c5 - c4; t[7] = c6 - c5; t[8] = c7 - c6; t[9] = c8 - c7; }
    out[threadIdx.x] = x + acc[0] + b0[1] + b1[2] + b2[3] + b3[0] + r;
}
int main() {
    double* out; long long* t;
    hipMalloc(&out, 8 * 256); hipMalloc(&t, 8 * 16);
    for (int wgs : {1, 256}) for (int rep = 0; rep < 3; ++rep) {
        const int n = 64;
        hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, out, t, n);
        long long h[16]; hipMemcpy(h, t, 8 * 16, hipMemcpyDeviceToHost);
        const double mhz = (double)h[1] / ((double)h[0] / 100.0), ops = 16.0 * n;
        printf("wgs %3d rep %d: kernel %.1f us, clock64 rate %.0f MHz | cycles per op: dep fma %.1f, indep fma %.1f, readlane+fma %.1f, lds+fma %.1f, dep mfma %.1f, indep mfma %.1f, rsq+add %.1f, barrier %.1f\n",
               wgs, rep, h[0] / 100.0, mhz, h[2] / ops, h[3] / ops, h[4] / ops, h[5] / ops, h[6] / ops, h[7] / ops, h[8] / ops, h[9] / (double)n);
    }
    return 0;
}
