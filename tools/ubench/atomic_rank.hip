// atomic_rank.hip -- cost of ranking N keys inside their bins with returning device-scope atomics (counting sort, first pass):
//     rank[i] = atomicAdd(&count[key[i]], 1)
// for (a) spatially coherent keys (runs of 8 consecutive equal keys, bins ascending), (b) the same bins in random order, (c) uniformly random keys.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/atomic_rank.hip -o tools/ubench/atomic_rank
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(256) void k_rank(const uint32_t* __restrict__ key, uint32_t n, uint32_t* __restrict__ count, uint32_t* __restrict__ rank) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) rank[i] = atomicAdd(&count[key[i]], 1u);
}
// the same with one atomic per run of equal keys among consecutive lanes (leader adds the run length, the others take their offset)
__global__ __launch_bounds__(256) void k_rank_runs(const uint32_t* __restrict__ key, uint32_t n, uint32_t* __restrict__ count, uint32_t* __restrict__ rank) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const uint32_t k = i < n ? key[i] : 0xFFFFFFFFu;
    const uint32_t prev = __shfl_up(k, 1);
    const bool head = lane == 0 || prev != k;
    const unsigned long long heads = __ballot(head);
    const unsigned long long below = heads & ((2ull << lane) - 1ull);          // heads at or below this lane
    const int my_head = 63 - __clzll(below);
    const unsigned long long above = heads & ~((2ull << lane) - 1ull);         // heads above this lane
    const int next_head = above ? __ffsll((long long)above) - 1 : 64;
    uint32_t base = 0;
    if (head && i < n) {
        const unsigned long long rest = heads & ~((2ull << lane) - 1ull);
        const int end = rest ? __ffsll((long long)rest) - 1 : 64;
        base = atomicAdd(&count[k], (uint32_t)(end - lane));
    }
    (void)next_head;
    base = __shfl(base, my_head);
    if (i < n) rank[i] = base + (uint32_t)(lane - my_head);
}
int main() {
    const uint32_t n = 10000000u, bins = 4700000u;
    std::vector<uint32_t> h(n);
    uint32_t *dk, *dc, *dr;
    hipMalloc(&dk, n * 4); hipMalloc(&dc, bins * 4); hipMalloc(&dr, n * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 3; ++mode) {
        if (mode == 0) for (uint32_t i = 0; i < n; ++i) h[i] = (uint32_t)(((uint64_t)(i / 8) * 3u) % bins);
        if (mode == 1) { std::vector<uint32_t> perm(n / 8 + 1); for (size_t j = 0; j < perm.size(); ++j) perm[j] = (uint32_t)((j * 2654435761ull) % bins); for (uint32_t i = 0; i < n; ++i) h[i] = perm[i / 8]; }
        if (mode == 2) { srand(1); for (uint32_t i = 0; i < n; ++i) h[i] = (uint32_t)(((uint64_t)rand() * 2654435761ull + (uint64_t)rand()) % bins); }
        hipMemcpy(dk, h.data(), n * 4, hipMemcpyHostToDevice);
        for (int variant = 0; variant < 2; ++variant) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemsetAsync(dc, 0, bins * 4, 0);
                hipEventRecord(a, 0);
                if (variant == 0) hipLaunchKernelGGL(k_rank, dim3((n + 255) / 256), dim3(256), 0, 0, dk, n, dc, dr);
                else hipLaunchKernelGGL(k_rank_runs, dim3((n + 255) / 256), dim3(256), 0, 0, dk, n, dc, dr);
                hipEventRecord(b, 0); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
            }
            printf("keys %-28s %-22s %.3f ms for %u keys\n", mode == 0 ? "runs of 8, ascending bins" : (mode == 1 ? "runs of 8, scattered bins" : "uniformly random"),
                   variant == 0 ? "one atomic per key" : "one atomic per run", best, n);
        }
    }
    return 0;
}
