// Throughput of returnless device-scope atomicAdd on global memory, MI355X: N atomics from the whole chip onto an array
// of `words` counters (random addresses), vs plain stores of the same pattern.
//   hipcc --offload-arch=gfx950 -O3 atomic_rate.hip -o atomic_rate && ./atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k_atomic(uint32_t* a, uint32_t words_mask, uint32_t per_thread, int mode) {
    uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    for (uint32_t i = 0; i < per_thread; ++i) {
        x = x * 1664525u + 1013904223u;
        uint32_t idx = (x >> 8) & words_mask;
        if (mode == 1) idx = (idx & ~255u) | (threadIdx.x & 255u);   // 256 consecutive words per workgroup-ish (row pattern)
        if (mode == 2) a[idx] = x;  // plain store, same addresses
        else atomicAdd(&a[idx], 1u);
    }
}
int main() {
    uint32_t* d;
    const size_t max_words = 1u << 24;
    hipMalloc(&d, max_words * 4);
    hipMemset(d, 0, max_words * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode)
        for (uint32_t words_log2 : {10u, 14u, 18u, 22u}) {
            const uint32_t blocks = 2048, threads = 256, per = 16;  // 8.4 M atomics
            k_atomic<<<blocks, threads>>>(d, (1u << words_log2) - 1u, per, mode);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 5; ++r) k_atomic<<<blocks, threads>>>(d, (1u << words_log2) - 1u, per, mode);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double n = 5.0 * blocks * threads * per;
            printf("mode %d (%s) words 2^%u: %.1f G ops/s (%.3f ms per 8.4 M)\n", mode,
                   mode == 0 ? "atomic random" : (mode == 1 ? "atomic row-coalesced" : "store random"), words_log2,
                   n / (ms * 1e-3) / 1e9, ms / 5);
        }
    return 0;
}
