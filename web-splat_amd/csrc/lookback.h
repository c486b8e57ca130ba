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
// Order of the chain.  K1 and the binning prefix use blockIdx order: a workgroup publishes its aggregate BEFORE it
// looks back, and the hardware dispatches the workgroups of a grid in increasing blockIdx order on every XCD, so
// whatever a workgroup waits for is running or about to be dispatched on an XCD whose slots are held only by
// workgroups that make progress -- no cycle.  (The same assumption CUB's decoupled look-back makes.  Measured on
// MI355X: handing the ids out by an atomic ticket instead -- start order, no assumption -- serialises ~11 ns per
// workgroup on one address: 56 -> 54 us in K1 at 1172 workgroups, 134 -> 109 us in the binning prefix at 2441;
// -DWS_TICKET_ORDER selects it.)  The one-sweep sort's tiles are handed out by ticket.  Every spin is bounded and
// reports through an error word instead of hanging the GPU, whatever the order.
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
