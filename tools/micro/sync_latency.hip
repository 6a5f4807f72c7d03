// Host round trip after a tiny kernel: hipStreamSynchronize against polling a sequence number the
// kernel writes into mapped host memory.  Build: hipcc --offload-arch=gfx950 -O2 sync_latency.hip -o sync_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_work(double* buf, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) buf[i] = buf[i] * 1.0000001 + 1.0; }
__global__ void k_pub(volatile int* h, int seq) { if (threadIdx.x == 0) { __threadfence_system(); h[0] = seq; } }
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    double* buf; hipMalloc(&buf, 8 << 20); hipMemset(buf, 0, 8 << 20);
    int* h; hipHostMalloc((void**)&h, 64, hipHostMallocMapped); h[0] = 0;
    const int reps = 2000;
    for (int mode = 0; mode < 2; ++mode) {
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 1; r <= reps; ++r) {
            hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, s, buf, 65536);
            hipLaunchKernelGGL(k_pub, dim3(1), dim3(64), 0, s, h, mode * reps + r);
            if (mode == 0) hipStreamSynchronize(s);
            else { while (*(volatile int*)h != mode * reps + r) { } }
        }
        hipStreamSynchronize(s);
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        printf("%s: %.2f us per {2 launches + wait}\n", mode == 0 ? "hipStreamSynchronize" : "poll mapped host word", us);
    }
    return 0;
}
