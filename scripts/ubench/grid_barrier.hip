// grid_barrier.hip -- what does a device-wide barrier INSIDE a launch cost on MI355X, against the dependent kernel
// boundary it would replace?  (round-3 verdict item 1: "a negative result counts only with the barrier cost measured in
// isolation".)  The barrier is the library's own (web-splat_amd/csrc/experimental/grid_barrier.h), the phases around it are the same
// in both forms:
//   phase   every workgroup reads the word its LEFT neighbour (blockIdx - 1) wrote in the previous phase, adds one and
//           writes its own word -- a real cross-workgroup dependency, so a barrier / boundary that does not order memory
//           shows up as a wrong final value -- and optionally streams `payload` bytes of stores (dirty L2 lines that the
//           release has to write back: a sort pass leaves tens of KB per workgroup).
//   chain   N launches of one phase each on one stream (the "dependent kernel boundary")
//   fused   ONE launch of N phases with gb::sync between them
// Prints one JSON object per configuration: us per phase for both forms and their difference (= barrier - boundary).
//   hipcc --offload-arch=gfx950 -O3 -I web-splat_amd/csrc scripts/ubench/grid_barrier.hip -o grid_barrier && ./grid_barrier
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "experimental/grid_barrier.h"

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));            \
            std::exit(1);                                                           \
        }                                                                           \
    } while (0)

__device__ __forceinline__ void phase_body(uint32_t* cur, const uint32_t* prev, uint32_t* slab, uint32_t payload_words,
                                           uint32_t phase) {
    const uint32_t b = blockIdx.x, g = gridDim.x;
    // stream the payload first (16-B stores, the whole workgroup)
    uint4* s4 = reinterpret_cast<uint4*>(slab + (size_t)b * payload_words);
    for (uint32_t i = threadIdx.x; i < payload_words / 4u; i += blockDim.x) s4[i] = make_uint4(phase, i, b, 0u);
    if (threadIdx.x == 0) {
        const uint32_t left = __hip_atomic_load(prev + ((b + g - 1u) % g) * 16u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cur[b * 16u] = left + 1u;  // a PLAIN store: only the barrier's / boundary's release makes it visible
    }
}

__global__ void k_phase(uint32_t* cur, const uint32_t* prev, uint32_t* slab, uint32_t payload_words, uint32_t phase) {
    phase_body(cur, prev, slab, payload_words, phase);
}

__global__ void k_fused(uint32_t* a, uint32_t* b, uint32_t* slab, uint32_t payload_words, uint32_t nphase, uint32_t* state,
                        uint32_t* err) {
    for (uint32_t p = 0; p < nphase; ++p) {
        uint32_t* cur = (p & 1u) ? b : a;
        const uint32_t* prev = (p & 1u) ? a : b;
        phase_body(cur, prev, slab, payload_words, p);
        if (p + 1u < nphase) ws::gb::sync(state, gridDim.x, p + 1u, err, 1u);
    }
}

__global__ void k_xcc(uint32_t* out) {
    if (threadIdx.x == 0) {
        uint32_t x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        out[blockIdx.x] = x & 15u;
    }
}

int main() {
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint32_t NPH = 64;
    uint32_t *a, *b, *slab, *state, *err;
    const size_t max_payload_words = 64u * 1024u / 4u;
    CK(hipMalloc(&a, 1024 * 64));
    CK(hipMalloc(&b, 1024 * 64));
    CK(hipMalloc(&slab, 1024 * max_payload_words * 4));
    CK(hipMalloc(&state, ws::gb::STATE_WORDS * 4));
    CK(hipMalloc(&err, 4));
    // where do consecutive workgroups land?  (the barrier groups by blockIdx & 7; k_sort's tiles assume b % 8 = XCD)
    {
        CK(hipMemsetAsync(a, 0xFF, 1024 * 4, st));  // (stream-ordered: `st` does not synchronise with the NULL stream)
        hipLaunchKernelGGL(k_xcc, dim3(256), dim3(256), 0, st, a);
        CK(hipStreamSynchronize(st));
        std::vector<uint32_t> h(256);
        CK(hipMemcpy(h.data(), a, 256 * 4, hipMemcpyDeviceToHost));
        int match = 0;
        for (int i = 0; i < 256; ++i) match += (h[i] == (uint32_t)(i & 7));
        std::printf("{\"probe\": \"xcc_id of workgroup b equals b %% 8\", \"workgroups\": 256, \"matching\": %d}\n", match);
    }
    const uint32_t grids[] = {64, 128, 256, 512};
    const uint32_t threads[] = {256, 1024};
    const uint32_t payloads[] = {0, 16 * 1024, 64 * 1024};
    for (uint32_t T : threads)
        for (uint32_t G : grids) {
            if (G * T > 256u * 2048u) continue;  // all workgroups resident (2048 threads per CU)
            for (uint32_t P : payloads) {
                const uint32_t pw = P / 4u;
                float best_chain = 1e30f, best_fused = 1e30f;
                bool ok = true;
                for (int rep = 0; rep < 5; ++rep) {
                    CK(hipMemsetAsync(a, 0, 1024 * 64, st));
                    CK(hipMemsetAsync(b, 0, 1024 * 64, st));
                    CK(hipEventRecord(e0, st));
                    for (uint32_t p = 0; p < NPH; ++p)
                        hipLaunchKernelGGL(k_phase, dim3(G), dim3(T), 0, st, (p & 1u) ? b : a, (p & 1u) ? a : b, slab, pw, p);
                    CK(hipEventRecord(e1, st));
                    CK(hipStreamSynchronize(st));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best_chain) best_chain = ms;
                    std::vector<uint32_t> h(G * 16);
                    CK(hipMemcpy(h.data(), ((NPH - 1) & 1u) ? b : a, G * 64, hipMemcpyDeviceToHost));
                    // after NPH phases every word has been incremented NPH times along the ring
                    for (uint32_t i = 0; i < G; ++i) ok = ok && (h[i * 16] == NPH);

                    CK(hipMemsetAsync(a, 0, 1024 * 64, st));
                    CK(hipMemsetAsync(b, 0, 1024 * 64, st));
                    CK(hipMemsetAsync(state, 0, ws::gb::STATE_WORDS * 4, st));
                    CK(hipMemsetAsync(err, 0, 4, st));
                    CK(hipEventRecord(e0, st));
                    hipLaunchKernelGGL(k_fused, dim3(G), dim3(T), 0, st, a, b, slab, pw, NPH, state, err);
                    CK(hipEventRecord(e1, st));
                    CK(hipStreamSynchronize(st));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best_fused) best_fused = ms;
                    CK(hipMemcpy(h.data(), ((NPH - 1) & 1u) ? b : a, G * 64, hipMemcpyDeviceToHost));
                    for (uint32_t i = 0; i < G; ++i) ok = ok && (h[i * 16] == NPH);
                    uint32_t herr = 0;
                    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
                    ok = ok && herr == 0;
                }
                std::printf("{\"workgroups\": %u, \"threads\": %u, \"payload_bytes_per_workgroup\": %u, \"phases\": %u, "
                            "\"chain_us_per_phase\": %.3f, \"fused_us_per_phase\": %.3f, \"barrier_minus_boundary_us\": %.3f, "
                            "\"values_correct\": %s}\n",
                            G, T, P, NPH, best_chain * 1e3f / NPH, best_fused * 1e3f / NPH,
                            (best_fused - best_chain) * 1e3f / NPH, ok ? "true" : "false");
                std::fflush(stdout);
            }
        }
    return 0;
}
