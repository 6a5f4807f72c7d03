// Scratch probe: HIP virtual memory management on the box -- reserve a large range, back only a window of it, touch the window.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_fill(double* p, size_t n, double v) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
int main() {
    int dev = 0; CK(hipSetDevice(dev));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    printf("granularity %zu\n", gran);
    const size_t total = (size_t)64 << 30;                          // 64 GB of address space
    void* base = nullptr; CK(hipMemAddressReserve(&base, total, 0, nullptr, 0));
    const size_t win = 64 * gran, off = 1000 * gran;
    hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, win, &prop, 0));
    CK(hipMemMap((char*)base + off, win, 0, h, 0));
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess((char*)base + off, win, &acc, 1));
    double* p = (double*)((char*)base + off);
    k_fill<<<(unsigned)((win / 8 + 255) / 256), 256>>>(p, win / 8, 3.5);
    CK(hipDeviceSynchronize());
    double v = 0; CK(hipMemcpy(&v, p + win / 8 - 1, 8, hipMemcpyDeviceToHost));
    size_t fr = 0, tot = 0; CK(hipMemGetInfo(&fr, &tot));
    printf("mapped %zu MB of a %zu GB reservation, last element %.1f, device free %.1f GB\n", win >> 20, total >> 30, v, fr / 1e9);
    CK(hipMemUnmap((char*)base + off, win)); CK(hipMemRelease(h)); CK(hipMemAddressFree(base, total));
    printf("ok\n");
    return 0;
}
