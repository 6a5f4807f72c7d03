// The row kernels (k_lin_plain, k_spmv_f) gather 3-vectors of fp64 from a staged tile + halo in LDS by neighbour id: rocprofv3 shows
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE ~ 0.5 for both.  Is that the [row][3] layout (24-byte stride), or the gather?
// Each lane gathers the vectors of pseudo-random rows of a 584-row tile (the C4 tile + halo size) from four layouts:
//   S3  [row][3]   stride 24 B (the product's)          S4  [row][4]   stride 32 B (one b128 + one b64 per vector)
//   S5  [row][5]   stride 40 B                          SOA x[] y[] z[] planes, stride 8 B
// and, as the conflict-free bound, consecutive rows (lane l reads row base + l).  Clock cycles per wave-level 3-vector gather.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/lds_gather_probe.hip -o /tmp/ldsg && /tmp/ldsg
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int ROWS = 584, ITERS = 2048;

template <int MODE, bool RANDOM>   // MODE 3 / 4 / 5: row stride in doubles; 0: SoA planes
__global__ __launch_bounds__(256, 4) void k(const int* __restrict__ ids, double* out, long long* clk) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    constexpr int N = MODE ? MODE * ROWS : 3 * ROWS;
    for (int i = tid; i < N; i += 256) lds[i] = (double)(i % 97);
    __syncthreads();
    double a0 = 0, a1 = 0, a2 = 0;
    const long long t0 = wall_clock64();
    const long long c0 = clock64();
    for (int it = 0; it < ITERS; it += 4) {
        int o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = RANDOM ? ids[(size_t)(it + q) * 256 + tid] : ((it + q) * 7 + tid) % ROWS;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (MODE) { a0 += lds[MODE * o[q]]; a1 += lds[MODE * o[q] + 1]; a2 += lds[MODE * o[q] + 2]; }
            else { a0 += lds[o[q]]; a1 += lds[ROWS + o[q]]; a2 += lds[2 * ROWS + o[q]]; }
        }
    }
    const long long c1 = clock64();
    const long long t1 = wall_clock64();
    out[(size_t)blockIdx.x * 256 + tid] = a0 + a1 + a2;
    if (tid == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = t1 - t0; }
}

template <int MODE, bool RANDOM>
static void run(const char* name, const int* d_ids, double* d_out, long long* d_clk, int nblk) {
    const size_t shm = sizeof(double) * (MODE ? MODE : 3) * ROWS;
    hipLaunchKernelGGL((k<MODE, RANDOM>), dim3(nblk), dim3(256), shm, 0, d_ids, d_out, d_clk);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<MODE, RANDOM>), dim3(nblk), dim3(256), shm, 0, d_ids, d_out, d_clk);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(2 * nblk);
    hipMemcpy(h.data(), d_clk, sizeof(long long) * 2 * nblk, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int b = 0; b < nblk; ++b) { cyc += (double)h[2 * b]; wall += (double)h[2 * b + 1]; }
    printf("%-34s %7.1f shader clocks, %6.2f ns (100 MHz wall clock) per wave-level 3-vector gather; launch %.1f us\n", name, cyc / nblk / ITERS, wall / nblk / ITERS * 10.0,
           1e3 * ms / 10);
}

int main() {
    const int nblk = 1024;                                         // 4 workgroups per CU, as the row kernels run
    std::vector<int> ids((size_t)ITERS * 256);
    unsigned s = 12345u;
    for (auto& v : ids) { s = s * 1664525u + 1013904223u; v = (int)((s >> 8) % ROWS); }
    int* d_ids; double* d_out; long long* d_clk;
    hipMalloc(&d_ids, sizeof(int) * ids.size()); hipMalloc(&d_out, sizeof(double) * 256 * nblk); hipMalloc(&d_clk, sizeof(long long) * 2 * nblk);
    hipMemcpy(d_ids, ids.data(), sizeof(int) * ids.size(), hipMemcpyHostToDevice);
    run<3, true>("random rows, [row][3] (24 B)", d_ids, d_out, d_clk, nblk);
    run<4, true>("random rows, [row][4] (32 B)", d_ids, d_out, d_clk, nblk);
    run<5, true>("random rows, [row][5] (40 B)", d_ids, d_out, d_clk, nblk);
    run<0, true>("random rows, x[] y[] z[] planes", d_ids, d_out, d_clk, nblk);
    run<3, false>("consecutive rows, [row][3]", d_ids, d_out, d_clk, nblk);
    run<4, false>("consecutive rows, [row][4]", d_ids, d_out, d_clk, nblk);
    run<0, false>("consecutive rows, planes", d_ids, d_out, d_clk, nblk);
    return 0;
}
