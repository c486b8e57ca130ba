// grid_barrier.h -- a device-wide barrier for kernels whose workgroups are ALL resident (grid <= what the chip holds).
//
// Used by the single-launch form of the depth sort (sort.hip, k_dsort_fat<.., COOP>) and priced in isolation by
// scripts/ubench/grid_barrier.hip against the dependent kernel boundary it replaces (round-3 verdict, item 1).
//
// Two levels, after MI355X_MICROARCH.md "barrier-xcd": the workgroups are split into 8 groups by blockIdx & 7 (workgroup b
// runs on XCD b % 8 -- observed; the grouping is for speed only, nothing below depends on placement), each group counts its
// arrivals on its own 64-B line, the LAST arriver of a group bumps the top counter, waits until all groups have, and
// publishes the group's generation; the other members poll their group's generation word.  256 pollers on one line
// become 8 x 32 on eight lines plus 8 on the ninth.
//
// Memory: EVERY workgroup releases its own writes (agent scope: write back this XCD's dirty L2 lines) before it arrives
// and acquires (invalidate L1 / non-coherent L2 lines) after it leaves -- the per-XCD L2s of MI355X are not coherent with
// each other, and no assumption is made about which XCD a workgroup's stores sit in.  The counters are monotonic: the
// state is zeroed ONCE (per frame, with the frame's zero arena) and the k-th barrier of the launch waits for k x members.
//
// Every spin is bounded; a time-out sets `error_bit` in *error_word and lets the workgroup proceed (wrong result, flagged,
// instead of a hung GPU).
//
// EXPERIMENTAL BUILD ONLY (round 5: -DWS_EXPERIMENTAL; the product library does not contain this barrier or its one user).
// Hardware assumption, stated because the fences below are weaker than the formal model asks for: only thread 0
// of a workgroup issues the release / acquire fences, and a group's last arriver bumps the top counter without an acquire of
// its group line.  That is sufficient on gfx9 parts -- buffer_wbl2 / buffer_inv act on the whole CU / XCD, and __syncthreads()
// drains the workgroup's vmcnt in front of the release -- and it is validated on gfx950 only; the data exchanged between the
// passes are plain loads and stores.  Several single-launch sorts in flight starve each other of slots (each needs all its
// workgroups resident): one frame at a time only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ws {
namespace gb {

constexpr int GROUPS = 8;
constexpr int STRIDE = 16;                       // uint32 words per line (64 B)
constexpr int STATE_WORDS = (GROUPS + 1) * STRIDE;  // 8 group lines {arrivals, generation} + the top counter's line
constexpr uint32_t SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ uint32_t ld(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// k = 1, 2, 3, ...: index of this barrier within the launch (all workgroups pass the same sequence).
// grid = number of workgroups taking part (gridDim.x).  Called by ALL threads of every workgroup.
__device__ __forceinline__ void sync(uint32_t* state, uint32_t grid, uint32_t k, uint32_t* error_word, uint32_t error_bit) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t g = blockIdx.x & (GROUPS - 1);
        const uint32_t members = (grid + (GROUPS - 1) - g) / GROUPS;          // workgroups b < grid with b % 8 == g
        const uint32_t ngroups = grid < (uint32_t)GROUPS ? grid : (uint32_t)GROUPS;
        uint32_t* line = state + g * STRIDE;
        uint32_t* top = state + GROUPS * STRIDE;
        __atomic_thread_fence(__ATOMIC_RELEASE);  // agent scope by default for HIP device code: my workgroup's writes are visible
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the compiler may drop the wait behind the write-back: microarch guide)
        const uint32_t old = __hip_atomic_fetch_add(line, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t spins = 0;
        if (old + 1u == members * k) {  // the group's last arriver
            __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (ld(top) < ngroups * k) {
                if (++spins > SPIN_LIMIT) {
                    if (error_word) atomicOr(error_word, error_bit);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            st(line + 1, k);
        } else {
            while (ld(line + 1) < k) {
                if (++spins > SPIN_LIMIT) {
                    if (error_word) atomicOr(error_word, error_bit);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
}

}  // namespace gb
}  // namespace ws
