// Does the record layout matter to HBM?  The lineariser streams, per 64-slot chunk of a slice, spring headers (4 B + 4 B in, 8 B
// out per slot) and damper headers (8 B + 4 B in, 8 B out).  SoA: six arrays.  AoSoA: the fields of one chunk contiguous
// (1024 B / 1280 B blocks).  One wave walks `chunks` consecutive chunks of each stream, like a slice; same bytes either way.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/aosoa_probe.hip -o /tmp/aosoa && /tmp/aosoa [Mslots] [chunks per wave]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <bool BLOCKED>
__global__ __launch_bounds__(256) void k(const char* __restrict__ sp, char* __restrict__ spw, const char* __restrict__ dm, char* __restrict__ dmw,
                                         size_t n_chunks, int chunks) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t c0 = wave * chunks;
    if (c0 >= n_chunks) return;
    double acc = 0;
    for (int c = 0; c < chunks && c0 + c < n_chunks; ++c) {
        const size_t ch = c0 + c;
        uint32_t om; float d0; uint2 hdr; float w;
        if (BLOCKED) {
            const char* b = sp + ch * 1024;
            om = reinterpret_cast<const uint32_t*>(b)[lane];
            d0 = reinterpret_cast<const float*>(b + 256)[lane];
            const char* d = dm + ch * 1280;
            hdr = reinterpret_cast<const uint2*>(d)[lane];
            w = reinterpret_cast<const float*>(d + 512)[lane];
        } else {
            om = reinterpret_cast<const uint32_t*>(sp)[ch * 64 + lane];
            d0 = reinterpret_cast<const float*>(sp + n_chunks * 256)[ch * 64 + lane];
            hdr = reinterpret_cast<const uint2*>(dm)[ch * 64 + lane];
            w = reinterpret_cast<const float*>(dm + n_chunks * 512)[ch * 64 + lane];
        }
        const double q = (double)d0 * (double)(om & 0xFFFF) + 1.0, s = (double)w * (double)(hdr.x & 0xFFFF) + (double)(hdr.y >> 16);
        acc += q * s;
        if (BLOCKED) {
            reinterpret_cast<double*>(spw + ch * 1024 + 512)[lane] = q;
            reinterpret_cast<double*>(dmw + ch * 1280 + 768)[lane] = s;
        } else {
            reinterpret_cast<double*>(spw + n_chunks * 512)[ch * 64 + lane] = q;
            reinterpret_cast<double*>(dmw + n_chunks * 768)[ch * 64 + lane] = s;
        }
    }
    if (acc == 1.2345e-300) spw[0] = 1;
}

int main(int argc, char** argv) {
    const size_t mslots = argc > 1 ? atol(argv[1]) : 128;
    const int chunks = argc > 2 ? atoi(argv[2]) : 12;
    const size_t n_chunks = mslots * 1000000 / 64;
    char *sp, *dm;
    hipMalloc(&sp, n_chunks * 1024); hipMalloc(&dm, n_chunks * 1280);
    hipMemset(sp, 1, n_chunks * 1024); hipMemset(dm, 1, n_chunks * 1280);
    const size_t waves = (n_chunks + chunks - 1) / chunks;
    const unsigned grid = (unsigned)((waves + 3) / 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocked = 0; blocked < 2; ++blocked)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (blocked) hipLaunchKernelGGL(k<true>, dim3(grid), dim3(256), 0, 0, sp, sp, dm, dm, n_chunks, chunks);
            else hipLaunchKernelGGL(k<false>, dim3(grid), dim3(256), 0, 0, sp, sp, dm, dm, n_chunks, chunks);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)n_chunks * 64 * (8 + 8 + 12 + 8);
            printf("{\"layout\": \"%s\", \"Mslots\": %zu, \"chunks_per_wave\": %d, \"ms\": %.3f, \"GBps\": %.0f}\n", blocked ? "AoSoA" : "SoA", mslots, chunks, ms, bytes / ms / 1e6);
        }
    return 0;
}
