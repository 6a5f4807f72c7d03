// Probe for the single-XCD persistent PCG: workers elected on one XCD (HW_REG_XCC_ID), a counter barrier between rounds,
// plain stores + L1-bypassing loads of rows other workgroups wrote.  Prints us per round and the number of stale words.
//   hipcc --offload-arch=gfx950 -O2 tools/xcd_barrier_probe.hip -o /tmp/xcd_probe && /tmp/xcd_probe [workers] [rounds] [halo] [rows per worker]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ inline int xcc_id() {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15;
}
__device__ inline double ld_l2(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct Ctl { int target, ticket, arrive, err; };

__global__ __launch_bounds__(256) void k_probe(Ctl* ctl, double* va, double* vb, int workers, int rounds, int halo, int rows_per,
                                              unsigned long long* stale, int* xcc_seen, int one_xcd) {
    __shared__ int s_b;
    const int tid = threadIdx.x;
    if (tid == 0) {
        int b = -1;
        const int me = xcc_id() + 1;
        int t = me;
        if (one_xcd == 1) {
            const int old = atomicCAS(&ctl->target, 0, me);
            t = old == 0 ? me : old;
        }
        if (t == me) {
            b = atomicAdd(&ctl->ticket, 1);
            if (b >= workers) b = -1;
            else xcc_seen[b] = me - 1;
        }
        s_b = b;
    }
    __syncthreads();
    const int b = s_b;
    if (b < 0) return;
    const int n_rows = workers * rows_per;
    unsigned long long bad = 0;
    for (int r = 0; r < rounds; ++r) {
        double* wr = (r & 1) ? vb : va;
        const double* rd = (r & 1) ? va : vb;
        // read rows other workers wrote in the previous round
        if (r > 0) {
            for (int i = tid; i < halo; i += 256) {
                const int row = (b * rows_per + rows_per + (i * 7) % (n_rows - rows_per)) % n_rows;
                for (int k = 0; k < 3; ++k) {
                    const double v = ld_l2(rd + 3 * (size_t)row + k);
                    if (v != (double)(r - 1) * 1000003.0 + 3 * row + k) ++bad;
                }
            }
        }
        if (tid < rows_per) {
            const int row = b * rows_per + tid;
            for (int k = 0; k < 3; ++k) {
                const double v = (double)r * 1000003.0 + 3 * row + k;
                if (one_xcd == 0) __hip_atomic_store(&wr[3 * (size_t)row + k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1: write-through
                else wr[3 * (size_t)row + k] = v;                                                                            // plain: stays in this XCD's L2
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(&ctl->arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int want = (r + 1) * workers;
            long long t0 = wall_clock64();
            while (__hip_atomic_load(&ctl->arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (wall_clock64() - t0 > 20000000LL) { ctl->err = 1; break; }   // 0.2 s at 100 MHz
            }
        }
        __syncthreads();
        if (__hip_atomic_load(&ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
    if (bad) atomicAdd(stale, bad);
}

int main(int argc, char** argv) {
    const int workers = argc > 1 ? atoi(argv[1]) : 35, rounds = argc > 2 ? atoi(argv[2]) : 2000, halo = argc > 3 ? atoi(argv[3]) : 520;
    const int rows_per = argc > 4 ? atoi(argv[4]) : 128;
    Ctl* ctl; double *va, *vb; unsigned long long* stale; int* seen;
    hipMalloc(&ctl, sizeof(Ctl)); hipMalloc(&va, 24 * (size_t)workers * rows_per); hipMalloc(&vb, 24 * (size_t)workers * rows_per);
    hipMalloc(&stale, 8); hipMalloc(&seen, 4 * workers);
    for (int one = 2; one >= 0; --one) {     // 2: any XCD, plain stores (stale by design); 1: one XCD, plain stores; 0: any XCD, sc1 stores
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(ctl, 0, sizeof(Ctl)); hipMemset(stale, 0, 8); hipMemset(seen, 0xff, 4 * workers);
            hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(k_probe, dim3(one == 1 ? 8 * (workers + 2) : workers), dim3(256), 0, 0, ctl, va, vb, workers, rounds, halo, rows_per, stale, seen, one);
            hipDeviceSynchronize();
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            Ctl h; unsigned long long hs; std::vector<int> hx(workers);
            hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost); hipMemcpy(&hs, stale, 8, hipMemcpyDeviceToHost);
            hipMemcpy(hx.data(), seen, 4 * workers, hipMemcpyDeviceToHost);
            int distinct = 0; for (int x = 0; x < 8; ++x) { bool any = false; for (int v : hx) any |= v == x; distinct += any; }
            printf("{\"one_xcd\": %d, \"workers\": %d, \"rounds\": %d, \"halo\": %d, \"us_per_round\": %.3f, \"stale_words\": %llu, \"timeout\": %d, \"ticket\": %d, \"xcds_used\": %d}\n",
                   one, workers, rounds, halo, us / rounds, hs, h.err, h.ticket, distinct);
        }
    }
    return 0;
}
