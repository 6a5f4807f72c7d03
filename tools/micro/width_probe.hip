// Does the per-lane ACCESS WIDTH of the incidence streams matter to HBM?  The lineariser streams, per slot, two 4-byte words in
// (header, static float) and one 8-byte factor out.  In the sliced-ELL order of the engine, slot (step j, lane l) of a slice sits
// at j * 64 + l: every load is a dword per lane (256 B per wave instruction), every store a dwordx2.  "Wide" order: slot (j, l) at
// (j / W) * 64 W + l * W + j % W, so a lane owns W consecutive words: dwordx4 loads (1 KB per wave instruction) and, for the
// factors, dwordx4 stores.  Same bytes either way; one wave walks `steps` steps of its slice, NB = 4 slots per lane in flight.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/width_probe.hip -o /tmp/width && /tmp/width [Mslots] [steps per slice]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int W, bool WR>
__global__ __launch_bounds__(256, 4) void k(const uint32_t* __restrict__ hdr, const float* __restrict__ st, double* __restrict__ fac, size_t n_slices, int steps) {
    const int lane = threadIdx.x & 63;
    const size_t slice = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slice >= n_slices) return;
    const size_t base = slice * (size_t)steps * 64;
    double acc = 0;
    for (int j = 0; j < steps; j += 4) {
        uint32_t h[4]; float s[4];
        if (W == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { h[q] = hdr[base + (size_t)(j + q) * 64 + lane]; s[q] = st[base + (size_t)(j + q) * 64 + lane]; }
        } else {
            const uint4 hv = *reinterpret_cast<const uint4*>(hdr + base + (size_t)j * 64 + lane * 4);
            const float4 sv = *reinterpret_cast<const float4*>(st + base + (size_t)j * 64 + lane * 4);
            h[0] = hv.x; h[1] = hv.y; h[2] = hv.z; h[3] = hv.w; s[0] = sv.x; s[1] = sv.y; s[2] = sv.z; s[3] = sv.w;
        }
        double f[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { f[q] = (double)s[q] * (double)(h[q] & 0xFFFF) + 1.0; acc += f[q]; }
        if (WR) {
            if (W == 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) fac[base + (size_t)(j + q) * 64 + lane] = f[q];
            } else {
                double2* o = reinterpret_cast<double2*>(fac + base + (size_t)j * 64 + lane * 4);
                o[0] = make_double2(f[0], f[1]); o[1] = make_double2(f[2], f[3]);
            }
        }
    }
    if (acc == 1.2345e-300) fac[0] = 1;
}

int main(int argc, char** argv) {
    const size_t mslots = argc > 1 ? atol(argv[1]) : 256;
    const int steps = argc > 2 ? atoi(argv[2]) : 12;
    const size_t n_slices = mslots * 1000000 / (64 * (size_t)steps);
    const size_t n = n_slices * steps * 64;
    uint32_t* hdr; float* st; double* fac;
    hipMalloc(&hdr, n * 4); hipMalloc(&st, n * 4); hipMalloc(&fac, n * 8);
    hipMemset(hdr, 1, n * 4); hipMemset(st, 1, n * 4); hipMemset(fac, 0, n * 8);
    const unsigned grid = (unsigned)((n_slices + 3) / 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wr = 1; wr >= 0; --wr)
        for (int wide = 0; wide < 2; ++wide)
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (wide && wr) hipLaunchKernelGGL((k<4, true>), dim3(grid), dim3(256), 0, 0, hdr, st, fac, n_slices, steps);
                else if (wide) hipLaunchKernelGGL((k<4, false>), dim3(grid), dim3(256), 0, 0, hdr, st, fac, n_slices, steps);
                else if (wr) hipLaunchKernelGGL((k<1, true>), dim3(grid), dim3(256), 0, 0, hdr, st, fac, n_slices, steps);
                else hipLaunchKernelGGL((k<1, false>), dim3(grid), dim3(256), 0, 0, hdr, st, fac, n_slices, steps);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double bytes = (double)n * (8 + (wr ? 8 : 0));
                printf("{\"order\": \"%s\", \"stores\": %d, \"Mslots\": %zu, \"steps\": %d, \"ms\": %.3f, \"GBps\": %.0f}\n", wide ? "wide4" : "dword", wr, mslots, steps, ms, bytes / ms / 1e6);
            }
    return 0;
}
