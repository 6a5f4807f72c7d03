// The 64 x 64 pivot-block sweep of the keyframe-block factorisation (csrc/nrs_kft_sweep.hpp) alone: result against a host inverse, cycles per sweep.
//   hipcc --offload-arch=gfx950 -O3 -I nr-slam_amd/csrc -I tools/micro tools/micro/sweep_blk_probe.hip -o /tmp/sweep_probe && /tmp/sweep_probe [scale] [real rows] [decay] [rank] [eps]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
namespace nrs {
typedef double nd_v4d __attribute__((ext_vector_type(4)));
template <int K>
__device__ inline double nd_rowbcast(double v) { return __builtin_amdgcn_update_dpp(v, v, 0x150 + K, 0xf, 0xf, false); }
template <int K>
__device__ inline void nd_fmacn_bcast(double& a, double ls, double l) {
    asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(ls), "v"(l), "n"(K));
}
constexpr int KFT_B = 64;
#include "nrs_kft_sweep.hpp"
#include "kft_sweep_chain_exp.hpp"
}  // namespace nrs
using namespace nrs;

__global__ __launch_bounds__(256) void k_probe(const double* A, double* out, long long* cyc, int* badf, int reps) {
    __shared__ double xb[KFT_SWEEP_XB];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    KftTiles c;
    long long t = 0;
    bool bad = false;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const double* a = A + (16 * w + (lane >> 4) + 4 * g) * 64 + (lane & 15);
            c.t0[g] = a[0]; c.t1[g] = a[16]; c.t2[g] = a[32]; c.t3[g] = a[48];
        }
        __syncthreads();
        const long long t0 = clock64();
        bad = kft_sweep64_blk(c, xb, lane, w) || bad;
        t += clock64() - t0;
        __syncthreads();
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        double* o = out + (16 * w + (lane >> 4) + 4 * g) * 64 + (lane & 15);
        o[0] = c.t0[g]; o[16] = c.t1[g]; o[32] = c.t2[g]; o[48] = c.t3[g];
    }
    if (tid == 0) { *cyc = t / reps; *badf = bad; }
}

__global__ __launch_bounds__(512) void k_probe_chain(const double* A, double* out, long long* cyc, int* badf, int reps) {
    __shared__ double xb[KFT_CHAIN_XB + 8];
    const int tid = threadIdx.x, lane = tid & 63, wphys = tid >> 6;
#ifdef CHAIN_ALONE                                                 // the chain wave alone on its SIMD: rows 0 on wave 5 (SIMD 1 holds two main waves), wave 0 idle
    const int w = wphys == 5 ? 0 : wphys == 0 ? 7 : wphys == 7 ? 5 : wphys;
#else
    const int w = wphys;
#endif
    KftTiles c;
    long long t = 0;
    int bad = 0;
    for (int r = 0; r < reps; ++r) {
        if (w < 4) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const double* a = A + (16 * w + (lane >> 4) + 4 * g) * 64 + (lane & 15);
                c.t0[g] = a[0]; c.t1[g] = a[16]; c.t2[g] = a[32]; c.t3[g] = a[48];
            }
        }
        __syncthreads();
        const long long t0 = clock64();
        kft_sweep64_chain(c, xb, lane, w);
        __syncthreads();
        t += clock64() - t0;
#ifdef KFT_EXP_STAMPS
        if (tid == 0 && r == reps - 1) {
            const long long* st = reinterpret_cast<const long long*>(xb + KFT_CHAIN_XB);
            printf("chain wave: entry +0, LDL start / end of the four blocks (cycles after entry):");
            for (int q = 0; q < 8; ++q) printf(" %lld", st[q] - t0);
            printf(", sweep end %lld\n", clock64() - t0);
        }
#endif
        bad |= kft_chain_bad(xb);
        __syncthreads();
    }
    if (w < 4) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            double* o = out + (16 * w + (lane >> 4) + 4 * g) * 64 + (lane & 15);
            o[0] = c.t0[g]; o[16] = c.t1[g]; o[32] = c.t2[g]; o[48] = c.t3[g];
        }
    }
    if (tid == 0) { *cyc = t / reps; *badf = bad; }
}

int main(int argc, char** argv) {
    const double scale = argc > 1 ? atof(argv[1]) : 1e6;
    const int n = 64, real = argc > 2 ? atoi(argv[2]) : 64;      // rows beyond `real` are identity pad rows (as the factorisation's blocks have them)
    const double decay = argc > 3 ? atof(argv[3]) : 0.0;          // row i scaled by 10^(-decay i / 64): an ill-conditioned block
    const int rank = argc > 4 ? atoi(argv[4]) : 200;             // X X^T of this rank ...
    const double eps_d = argc > 5 ? atof(argv[5]) : 1.0;         // ... + eps_d I: condition ~ 1 / eps_d for rank < 64
    std::vector<double> X(n * 200), A(n * n), inv(n * n);
    srand(7);
    for (auto& x : X) x = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0;
            for (int k = 0; k < rank; ++k) s += X[i * 200 + k] * X[j * 200 + k];
            A[i * n + j] = scale * (s + (i == j ? eps_d : 0.0)) * (1.0 + (i % 7)) * (1.0 + (j % 7)) * pow(10.0, -decay * ((i * 37) % 64) / 64.0) * pow(10.0, -decay * ((j * 37) % 64) / 64.0);
            if (i >= real || j >= real) A[i * n + j] = i == j ? 1.0 : 0.0;
        }
    // host inverse by Gauss-Jordan (long double)
    std::vector<long double> M(n * 2 * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) { M[i * 2 * n + j] = A[i * n + j]; M[i * 2 * n + n + j] = i == j; }
    for (int p = 0; p < n; ++p) {
        const long double d = M[p * 2 * n + p];
        for (int j = 0; j < 2 * n; ++j) M[p * 2 * n + j] /= d;
        for (int i = 0; i < n; ++i)
            if (i != p) {
                const long double f = M[i * 2 * n + p];
                for (int j = 0; j < 2 * n; ++j) M[i * 2 * n + j] -= f * M[p * 2 * n + j];
            }
    }
    // the same sweep pivot by pivot in double on the host: what rounding alone costs on this block
    std::vector<double> Hs(A);
    for (int p = 0; p < n; ++p) {
        const double d = Hs[p * n + p], iv = 1.0 / d;
        std::vector<double> col(n);
        for (int i = 0; i < n; ++i) col[i] = Hs[i * n + p];
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                if (i != p && j != p) Hs[i * n + j] -= col[i] * col[j] * iv;
        for (int i = 0; i < n; ++i) if (i != p) { Hs[i * n + p] = col[i] * iv; Hs[p * n + i] = col[i] * iv; }
        Hs[p * n + p] = -iv;
    }
    double herr = 0, hmx = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { hmx = fmax(hmx, fabs((double)M[i * 2 * n + n + j])); herr = fmax(herr, fabs(Hs[i * n + j] + (double)M[i * 2 * n + n + j])); }
    printf("host pivot-by-pivot sweep in double: %.3e\n", herr / hmx);
    double *dA, *dO; long long* dC; int* dB;
    hipMalloc(&dA, 8 * n * n); hipMalloc(&dO, 8 * n * n); hipMalloc(&dC, 8); hipMalloc(&dB, 4);
    hipMemcpy(dA, A.data(), 8 * n * n, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int pass = 0; pass < 2; ++pass) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(256), 0, 0, dA, dO, dC, dB, pass ? 2000 : 1);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%.2f us per {load block, sweep} (events over 2000 repetitions in one launch)\n", 1e3 * ms / 2000);
    std::vector<double> O(n * n);
    long long cyc; int bad;
    hipMemcpy(O.data(), dO, 8 * n * n, hipMemcpyDeviceToHost);
    hipMemcpy(&cyc, dC, 8, hipMemcpyDeviceToHost); hipMemcpy(&bad, dB, 4, hipMemcpyDeviceToHost);
    double err = 0, mx = 0; int wi = 0, wj = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const double r = (double)M[i * 2 * n + n + j];
            mx = fmax(mx, fabs(r));
            if (fabs(O[i * n + j] + r) > err) { err = fabs(O[i * n + j] + r); wi = i; wj = j; }
        }
    printf("sweep64_blk: max |c + A^-1| / max |A^-1| = %.3e (at %d, %d), bad %d, %lld clock64 ticks per sweep (s_memtime, 100 MHz)\n", err / mx, wi, wj, bad, cyc);
    {   // the chain-wave form (eight waves): bit-identical result expected
        std::vector<double> O2(n * n);
        for (int pass = 0; pass < 2; ++pass) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_probe_chain, dim3(1), dim3(512), 0, 0, dA, dO, dC, dB, pass ? 2000 : 1);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1);
        }
        hipMemcpy(O2.data(), dO, 8 * n * n, hipMemcpyDeviceToHost);
        hipMemcpy(&cyc, dC, 8, hipMemcpyDeviceToHost); hipMemcpy(&bad, dB, 4, hipMemcpyDeviceToHost);
        int ndiff = 0;
        for (int i = 0; i < n * n; ++i) ndiff += O2[i] != O[i];
        printf("sweep64_chain: %.2f us per {load block, sweep}, %lld ticks per sweep, bad %d, %d of %d entries differ from sweep64_blk\n", 1e3 * ms / 2000, cyc, bad, ndiff, n * n);
    }
    // per 16 x 16 tile error map
    for (int ti = 0; ti < 4; ++ti) {
        for (int tj = 0; tj < 4; ++tj) {
            double e = 0;
            for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) e = fmax(e, fabs(O[(16 * ti + i) * n + 16 * tj + j] + (double)M[(16 * ti + i) * 2 * n + n + 16 * tj + j]));
            printf(" %.1e", e / mx);
        }
        printf("\n");
    }
    return 0;
}
