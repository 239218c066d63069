// Does the HIP virtual-memory API work on this box, and what does a buffer made of N-MiB physical chunks cost a scattered-store kernel compared with
// hipMalloc?  (the "order effect" experiment)   hipcc --offload-arch=gfx950 -O3 vmm_probe.hip -o vmm_probe && ./vmm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void scatter(unsigned long long *p, size_t n, unsigned long long seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long x = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    __builtin_nontemporal_store(x, p + (x % n));
}
static float time_scatter(unsigned long long *p, size_t n) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(scatter, dim3(1 << 17), dim3(256), 0, 0, p, n, 1ull);
    hipEventRecord(a);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(scatter, dim3(1 << 17), dim3(256), 0, 0, p, n, 2ull + i);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main(int argc, char **argv) {
    const size_t bytes = (size_t)(argc > 1 ? atof(argv[1]) : 1.0) * (1ull << 30);
    int dev = 0; CK(hipSetDevice(0));
    void *m = nullptr; CK(hipMalloc(&m, bytes));
    printf("hipMalloc %zu MiB: scattered 8-byte stores %.3f ms per 33.5M\n", bytes >> 20, time_scatter((unsigned long long *)m, bytes / 8));
    hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    size_t gmin = 0, grec = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity: minimum %zu, recommended %zu\n", gmin, grec);
    for (size_t chunk_mb : {2, 64, 256}) {
        const size_t chunk = chunk_mb << 20;
        void *va = nullptr; CK(hipMemAddressReserve(&va, bytes, grec, nullptr, 0));
        std::vector<hipMemGenericAllocationHandle_t> hs;
        for (size_t off = 0; off < bytes; off += chunk) {
            hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, chunk, &prop, 0));
            CK(hipMemMap((char *)va + off, chunk, 0, h, 0)); hs.push_back(h);
        }
        hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(va, bytes, &acc, 1));
        CK(hipMemset(va, 0, bytes));
        printf("VMM, %zu-MiB chunks: scattered 8-byte stores %.3f ms per 33.5M\n", chunk_mb, time_scatter((unsigned long long *)va, bytes / 8));
        CK(hipDeviceSynchronize());
        size_t off = 0;
        for (auto h : hs) { CK(hipMemUnmap((char *)va + off, chunk)); CK(hipMemRelease(h)); off += chunk; }
        CK(hipMemAddressFree(va, bytes));
    }
    // churn: many allocations of odd sizes freed in a scattered order, then the same hipMalloc again
    std::vector<void *> junk;
    for (int i = 0; i < 400; i++) { void *q = nullptr; if (hipMalloc(&q, (size_t)(37 + (i * 7919) % 300) << 20) == hipSuccess) junk.push_back(q); }
    CK(hipFree(m));
    for (size_t i = 0; i < junk.size(); i += 2) hipFree(junk[i]);
    CK(hipMalloc(&m, bytes));
    printf("hipMalloc after churn: scattered 8-byte stores %.3f ms per 33.5M\n", time_scatter((unsigned long long *)m, bytes / 8));
    for (size_t i = 1; i < junk.size(); i += 2) hipFree(junk[i]);
    printf("done\n");
    return 0;
}
