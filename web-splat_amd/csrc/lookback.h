// lookback.h -- decoupled look-back primitives shared by the preprocess, binning and sort kernels.
//
// A status word is ONE naturally aligned 64-bit granule {epoch:32 | flag:2 | value:30}, written with a single
// agent-scope relaxed atomic store and polled with agent-scope relaxed atomic loads.  Because the value travels
// inside the flag's own word there is nothing to order: no release/acquire fence is needed, and the protocol is
// correct for any placement of producer and consumer on MI355X's 8 XCDs (non-coherent per-XCD L2s, per-CU L1s
// that are never refreshed by other CUs' stores) -- the "R2 granule" form of the CDNA hand-off rules.
//
// The epoch (a per-frame / per-sort counter passed as a kernel argument) replaces zeroing: a word whose epoch
// field differs from the current one is "not published yet", so the status arrays never need a memset
// (they are zeroed once at allocation; epoch 0 is never used).
//
// Order of the chain.  Work items are handed out by an atomic ticket, so a workgroup only ever waits on workgroups
// that have already started and hold their slot: no assumption about dispatch order or about what else is running.
// The ticket is one returning atomic on one address, ~11 ns each IN SERIES (3 us of K1's 45 at 1172 workgroups,
// 25 us of the binning prefix's 134 at 2441).  Taking the chain in blockIdx order instead removes that cost and is
// deadlock-free for ONE kernel (dispatch is in order per XCD) -- but not for several look-back kernels in flight:
// with four frames on four streams, workgroups of kernel A spin on XCDs whose remaining slots kernel B's missing
// predecessor needs while B's spinners hold the slots A's predecessor needs, and the cycle only breaks when an
// unrelated kernel drains.  Measured: 6.5 ms and 38 ms per frame instead of 0.18 and 0.35 ms (1 M Gaussians at
// 800x600, 2 M at 1920x1080).  Every spin is bounded and reports through an error word instead of hanging the GPU.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ws {
namespace lb {

constexpr uint32_t FLAG_AGG = 1u;   // the workgroup's own aggregate is available
constexpr uint32_t FLAG_INCL = 2u;  // the inclusive prefix up to and including this workgroup is available
constexpr uint32_t VALUE_MASK = (1u << 30) - 1u;
constexpr uint32_t SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ uint64_t pack(uint32_t epoch, uint32_t flag, uint32_t value) {
    return ((uint64_t)epoch << 32) | ((uint64_t)flag << 30) | (uint64_t)(value > VALUE_MASK ? VALUE_MASK : value);
}
__device__ __forceinline__ uint64_t ld(const uint64_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st(uint64_t* p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 0 = not published in this epoch
__device__ __forceinline__ uint32_t flag_of(uint64_t w, uint32_t epoch) {
    return ((uint32_t)(w >> 32) == epoch) ? (((uint32_t)w) >> 30) : 0u;
}
__device__ __forceinline__ uint32_t value_of(uint64_t w) { return ((uint32_t)w) & VALUE_MASK; }

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Exclusive prefix of one value per workgroup, for workgroup `bid`, computed by ONE wave: lane l inspects
// predecessor bid-1-l, so a poll covers 64 predecessors and the chain is at most bid/64 polls long even when
// every predecessor has only published its aggregate.
__device__ inline uint32_t wave_lookback(const uint64_t* status, uint32_t bid, uint32_t epoch, int lane,
                                         uint32_t* error_word, uint32_t error_bit) {
    uint32_t sum = 0;
    int64_t base = (int64_t)bid - 1;
    uint32_t spins = 0;
    while (true) {
        const int64_t i = base - lane;
        uint32_t flag = FLAG_INCL, val = 0u;  // virtual predecessor of workgroup 0: inclusive prefix 0
        if (i >= 0) {
            const uint64_t w = ld(status + i);
            flag = flag_of(w, epoch);
            val = value_of(w);
        }
        const unsigned long long incl = __ballot(flag == FLAG_INCL);
        const int first = incl ? (__ffsll((long long)incl) - 1) : 64;
        const unsigned long long pending = __ballot(flag == 0u && lane <= first);
        if (pending) {
            if (++spins > SPIN_LIMIT) {
                if (lane == 0 && error_word) atomicOr(error_word, error_bit);
                return sum;
            }
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        sum += wave_sum(lane <= first ? val : 0u);
        if (incl) return sum;
        base -= 64;
    }
}

}  // namespace lb
}  // namespace ws
