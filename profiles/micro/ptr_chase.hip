// Dependent-load latency on MI355X: every wave chases idx = A[idx].x through a random cyclic permutation of 16-byte
// records (the walk's node records) of a given footprint.  Build: hipcc --offload-arch=gfx950 -O3 ptr_chase.hip -o ptr_chase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

__global__ __launch_bounds__(64) void chase(const int4 *__restrict__ A, uint32_t n, uint32_t steps, uint32_t *out) {
    uint32_t idx = (uint32_t)(((uint64_t)blockIdx.x * 2654435761u) % n);
    for (uint32_t i = 0; i < steps; i++) idx = (uint32_t)A[idx].x;
    if (threadIdx.x == 0) out[blockIdx.x] = idx;
}

int main(int argc, char **argv) {
    const uint32_t steps = 20000;
    for (size_t mb : {1, 16, 64, 200, 1024, 4096}) {
        const size_t n = mb * (1 << 20) / 16;
        std::vector<uint32_t> perm(n);
        std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937_64 rng(1);
        std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<int4> h(n);
        for (size_t i = 0; i < n; i++) h[perm[i]] = make_int4((int)perm[(i + 1) % n], 0, 0, 0);
        int4 *d; uint32_t *o;
        hipMalloc(&d, n * 16); hipMalloc(&o, 1 << 20);
        hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
        for (uint32_t waves : {1u, 960u, 3840u, 15360u}) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipLaunchKernelGGL(chase, dim3(waves), dim3(64), 0, 0, d, (uint32_t)n, 100u, o);
            hipEventRecord(a);
            hipLaunchKernelGGL(chase, dim3(waves), dim3(64), 0, 0, d, (uint32_t)n, steps, o);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("footprint %6zu MB  waves %6u  %.0f ns per dependent load\n", mb, waves, ms * 1e6 / steps);
        }
        hipFree(d); hipFree(o);
    }
    return 0;
}
