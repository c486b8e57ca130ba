// sort.hip -- stable ascending LSD radix sort of (key, u32 value) pairs for gfx950.
//
// Replaces GPURSSorter (src/gpu_rs.rs:63-885) and src/shaders/radix_sort.wgsl:48-512 of the reference:
// same contract (ascending, stable, key count read from device memory), new design.
//
// Common to both paths (k_sort_scatter): one workgroup = one tile of 256 x KPT pairs; ranking inside a wave uses wave64
// ballots (one per digit bit) instead of the reference's O(subgroup) shared-memory match loop
// (radix_sort.wgsl:279-302); per-wave LDS digit counters are bumped by one leader lane per distinct digit, all keys'
// LDS atomics issued back to back (in order per wave: deterministic, stable); keys and values are reordered through
// LDS together so that global writes are contiguous per digit run; the first pass can synthesise the iota payload;
// tiles are dealt to workgroups so that one XCD owns a contiguous tile range (partial cache lines of neighbouring
// tiles' runs meet in one L2).  Digit width is a template parameter: 8 bits for 32-bit keys, 6..8 for the tile-id
// sort, whose keys may also be stored as 16-bit values.
//
// Cross-tile prefix ("tile histograms -> column scan -> scatter"): per pass a histogram kernel writes the digit counts of
// every tile (transposed, [digit][tile]), one workgroup per digit scans its row, the scatter kernel reads its offsets.  No
// spinning, nothing to order; costs one extra read of the keys per pass.  When the producer of the keys already knows the
// per-tile digit counts of the first pass (the tile binning kernel does) that histogram kernel is skipped.
// Two other forms were built, tested bit-exact and measured slower in two rounds each; they left the tree in round 4
// (git history): the classic one-sweep with a windowed decoupled look-back over 2048-pair tiles (every tile
// is resident and starts together on this chip: ~83 serial hops at 0.7 M keys) and the range-adaptive three-pass depth sort.
// The fat-tile one-sweep below (k_dsort_fat) is the form of that idea that does not chain; it is a measured variant too.
#include <hip/hip_runtime.h>

#ifdef WS_EXPERIMENTAL
#include "experimental/grid_barrier.h"  // (the single-launch depth sort's device-wide barrier)
#endif
#include "lookback.h"
#include "ws_internal.h"

namespace ws {

uint32_t sort_grid(uint32_t tiles);

namespace {

constexpr int WAVES = SORT_THREADS / 64;
constexpr int HIST_COPIES = 8;  // replicated LDS bins: lanes l and l+8k share a copy -> <= 8-way conflicts

__device__ __forceinline__ uint32_t device_count(const uint32_t* d_count, uint32_t n) {
    if (!d_count) return n;
    const uint32_t c = *d_count;
    return c < n ? c : n;
}

// Tile owned by linear work item L (= blockIdx.x + i * gridDim.x, gridDim.x a multiple of 8).  Workgroup b runs on
// XCD b % 8 (observed; used for speed only), so XCD x gets the CONTIGUOUS tile range [x * tpx, (x + 1) * tpx):
// the digit runs that neighbouring tiles write are adjacent in memory (a few dozen bytes each), and with both
// tiles on one XCD the partial cache lines meet in that XCD's L2 instead of being written back separately by
// two non-coherent L2s.  Returns false when the item has no tile.
__device__ __forceinline__ bool xcd_tile(uint32_t L, uint32_t ntiles, uint32_t* t) {
    const uint32_t tpx = (ntiles + 7u) >> 3;
    const uint32_t j = L >> 3;
    *t = (L & 7u) * tpx + j;
    return j < tpx && *t < ntiles;
}

// exclusive scan of one value per thread over a 256-thread block
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* s_tmp /*[WAVES]*/, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_tmp[wave] = incl;
    __syncthreads();
    uint32_t wave_off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        const uint32_t c = s_tmp[w];
        if (w < wave) wave_off += c;
        tot += c;
    }
    __syncthreads();  // s_tmp reusable
    if (total) *total = tot;
    return wave_off + incl - v;
}

// ---- digit counts of every tile, transposed: tile_sums[digit * tiles_cap + tile] -----------------------------------
template <int KPT, bool KEY16>
__global__ __launch_bounds__(SORT_THREADS) void k_sort_tile_hist(const uint32_t* __restrict__ keys,
                                                                const uint32_t* __restrict__ d_count, uint32_t n,
                                                                int shift, uint32_t mask, uint32_t* __restrict__ tile_sums,
                                                                uint32_t tiles_cap, FrameCounters* __restrict__ fold,
                                                                const uint32_t* __restrict__ skip,
                                                                const uint32_t* __restrict__ key_base) {
    WS_SETPRIO_SMALL();
    constexpr int TILE_N = SORT_THREADS * KPT;
    __shared__ uint32_t sh[RADIX * HIST_COPIES];
    __shared__ uint32_t s_rng[2][WAVES];
    // (base, skip and fold belong to the depth sort of a frame: 32-bit keys.  The 16-bit instantiations -- the tile-id sort --
    // compile to exactly what they were without them.)
    if (!KEY16 && skip && *skip) return;  // the depth sort's last pass over a constant digit (ws_internal.h depth_range_decide)
    const uint32_t kbase = (!KEY16 && key_base) ? *key_base : 0u;
    const uint32_t count = device_count(d_count, n);
    const uint32_t copy = threadIdx.x & (HIST_COPIES - 1);
    // the grid is capped (sort_grid): the host only knows the bound n, and workgroups that find nothing to do
    // still cost a dispatch slot each -- with n = 4 x count they made this kernel launch-rate bound
    const uint32_t ntiles = (count + TILE_N - 1) / TILE_N;
    for (uint32_t L = blockIdx.x; (L >> 3) < ((ntiles + 7u) >> 3); L += gridDim.x) {
        uint32_t t;
        if (!xcd_tile(L, ntiles, &t)) continue;  // block-uniform
        for (int i = threadIdx.x; i < RADIX * HIST_COPIES; i += SORT_THREADS) sh[i] = 0u;
        __syncthreads();
        const uint32_t base = t * TILE_N;
        uint32_t k[KPT];
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const uint32_t pos = base + j * SORT_THREADS + threadIdx.x;
            const uint32_t q = pos < count ? pos : count - 1u;
            k[j] = KEY16 ? (uint32_t)reinterpret_cast<const uint16_t*>(keys)[q] : keys[q];
        }
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const uint32_t pos = base + j * SORT_THREADS + threadIdx.x;
            if (pos < count) atomicAdd(&sh[(((k[j] - kbase) >> shift) & mask) * HIST_COPIES + copy], 1u);
        }
        if (!KEY16 && fold) {
            // The depth sort's first kernel reads every key anyway: it also leaves their range -- max(~key) and max(key), two
            // returnless atomics per tile into the 16 slotted lines of the frame counters -- for the column scan behind it to
            // fold into the base of passes 1..3 and the skip flag of pass 3 (ws_internal.h depth_range_decide).
            uint32_t knmin = 0u, kmax = 0u;
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                const uint32_t pos = base + j * SORT_THREADS + threadIdx.x;
                if (pos < count) {
                    knmin = max(knmin, ~k[j]);
                    kmax = max(kmax, k[j]);
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                knmin = max(knmin, (uint32_t)__shfl_xor((int)knmin, o, 64));
                kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, o, 64));
            }
            if ((threadIdx.x & 63) == 0) {
                s_rng[0][threadIdx.x >> 6] = knmin;
                s_rng[1][threadIdx.x >> 6] = kmax;
            }
        }
        __syncthreads();
        if (!KEY16 && fold && threadIdx.x == 0) {
            uint32_t knmin = 0u, kmax = 0u;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                knmin = max(knmin, s_rng[0][w]);
                kmax = max(kmax, s_rng[1][w]);
            }
            if (knmin) {
                uint32_t* ts = fold->tile_sums + (t & (TILE_SUM_SLOTS - 1)) * TILE_SUM_STRIDE;
                atomicMax(ts + 2, knmin);
                atomicMax(ts + 3, kmax);
            }
        }
        uint32_t c = 0;
#pragma unroll
        for (int r = 0; r < HIST_COPIES; ++r) c += sh[threadIdx.x * HIST_COPIES + r];
        if (threadIdx.x <= mask) tile_sums[(size_t)threadIdx.x * tiles_cap + t] = c;  // rows of the digits in use
        __syncthreads();
    }
}

// One workgroup per digit: exclusive scan of that digit's row over the tiles (in place), total -> hist[digit].
// Each thread owns a run of consecutive tiles: sum it, one block scan over the 256 run sums, then write the run's
// prefixes.  (A chunked loop with a block scan per 256 tiles cost 0.8 us per chunk in barriers: 10 us for the
// 3 k tiles of the tile-id sort, as much as the histogram kernel in front of it.)
__global__ __launch_bounds__(SORT_THREADS) void k_sort_col_scan(const uint32_t* __restrict__ d_count, uint32_t n,
                                                               uint32_t tile_n, uint32_t* __restrict__ tile_sums,
                                                               uint32_t tiles_cap, uint32_t* __restrict__ hist,
                                                               const uint32_t* __restrict__ skip, FrameCounters* __restrict__ fold) {
    WS_SETPRIO_SMALL();
    __shared__ uint32_t s_tmp[WAVES];
    if (skip && *skip) return;
    // the depth sort's first column scan: the key range the histogram kernel in front of it left -> the base of passes 1..3 and
    // ONE flag for the kernels of the last pass and the sort's readers
    if (fold && blockIdx.x == 0 && threadIdx.x == 0) depth_range_decide(fold, gridDim.x);  // (one workgroup per digit: the radix)
    const uint32_t count = device_count(d_count, n);
    const uint32_t ntiles = (count + tile_n - 1) / tile_n;
    uint32_t* row = tile_sums + (size_t)blockIdx.x * tiles_cap;
    const uint32_t per = (ntiles + SORT_THREADS - 1) / SORT_THREADS;  // run length, block-uniform
    const uint32_t t0 = threadIdx.x * per;
    const uint32_t t1 = (t0 + per < ntiles) ? t0 + per : ntiles;
    uint32_t sum = 0;
    for (uint32_t t = t0; t < t1; ++t) sum += row[t];
    uint32_t total;
    uint32_t run = block_exclusive_scan(sum, s_tmp, &total);
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t c = row[t];
        row[t] = run;
        run += c;
    }
    if (threadIdx.x == 0) hist[blockIdx.x] = total;
}

// ---- one digit pass: rank, precomputed cross-tile prefix, LDS reorder, scatter ------------------------------------
// RANGES (last pass of the tile-id sort only): the sorted keys themselves are never read again -- what the
// compositing pass needs is [begin, end) of every key value in the sorted order.  Equal keys of one workgroup are
// a contiguous run of its LDS-ordered tile, so run boundaries are found there and merged across workgroups with
// two atomicMax per (workgroup, key) pair on ranges[key] = (0xFFFFFFFF - begin, end), zero = empty.  This
// replaces a separate pass over the sorted keys and the final 4-B-per-entry key write.
// BITS: width of this sort's digits (8 for 32-bit keys; the tile-id sort splits its 12..16 key bits evenly over its
// passes, e.g. 6 + 6 for 3750 tiles: fewer ballots per key, shorter scans, longer write runs).
// CARRY: a 4-byte companion value (aux_in -> aux_out) travels with the payload: the depth sort carries the splat's packed
// tile rectangle, so that the binning prefix reads it in draw order instead of gathering it (raster.hip).
// KEY16: the key arrays hold uint16_t (tile ids below 65535): 2 B less per entry.  A template parameter, not an argument:
// with both load forms behind a run-time flag the compiler shared registers between them and put a full s_waitcnt vmcnt
// between the second and third key load of every thread -- two exposed round trips per tile instead of one.
template <int KPT, bool RANGES, int BITS, bool CARRY = false, bool KEY16 = false>
__global__ __launch_bounds__(SORT_THREADS) void k_sort_scatter(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, const uint32_t* __restrict__ aux_in, uint32_t* __restrict__ aux_out,
    const uint32_t* __restrict__ d_count, uint32_t n, int shift, int iota,
    const uint32_t* __restrict__ hist,     // [256] global digit histogram of this pass
    const uint32_t* __restrict__ tile_off, // [256][tiles_cap] exclusive offsets per digit
    uint32_t tiles_cap, uint2* __restrict__ ranges, uint32_t nranges, const uint32_t* __restrict__ skip,
    const uint32_t* __restrict__ key_base) {
  // key_base: digits come from (key - *key_base) (depth sort, passes 1..3)
    WS_SETPRIO_SMALL();
    constexpr int TILE_N = SORT_THREADS * KPT;
    constexpr uint32_t DMASK = (1u << BITS) - 1u;
    __shared__ uint32_t s_wave_hist[WAVES][RADIX];
    __shared__ uint32_t s_local_excl[RADIX];
    __shared__ uint32_t s_global_base[RADIX];
    __shared__ uint32_t s_keys[TILE_N];
    __shared__ uint32_t s_vals[TILE_N];
    __shared__ uint32_t s_aux[CARRY ? TILE_N : 1];
    __shared__ uint32_t s_tmp[WAVES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // (base and skip belong to the depth sort of a frame, the CARRY instantiation: every other one compiles to what it was)
    if (CARRY && skip && *skip) return;  // (block-uniform) a pass over a constant digit moves nothing
    const uint32_t kbase = (CARRY && key_base) ? *key_base : 0u;

    const uint32_t count = device_count(d_count, n);
    // first output position of every digit: the same for all tiles of this pass
    const uint32_t digit_base = block_exclusive_scan((uint32_t)tid <= DMASK ? hist[tid] : 0u, s_tmp, nullptr);
    // the grid is capped (sort_grid) and workgroups stride over the tiles
    const uint32_t ntiles = (count + TILE_N - 1) / TILE_N;
    for (uint32_t L = blockIdx.x; (L >> 3) < ((ntiles + 7u) >> 3); L += gridDim.x) {
    uint32_t t;
    if (!xcd_tile(L, ntiles, &t)) continue;  // block-uniform
    const uint32_t tile_base = t * TILE_N;
    const uint32_t valid = (count - tile_base) < (uint32_t)TILE_N ? (count - tile_base) : (uint32_t)TILE_N;

    // ---- load (wave-striped: consecutive lanes read consecutive keys; order = (wave, j, lane)) ------
    uint32_t key[KPT];
    uint32_t val[KPT];
    const uint32_t wave_base = tile_base + wave * (64 * KPT) + lane;
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const uint32_t pos = wave_base + j * 64;
        key[j] = pos < count ? (KEY16 ? (uint32_t)reinterpret_cast<const uint16_t*>(keys_in)[pos] : keys_in[pos])
                             : 0xFFFFFFFFu;
    }
    // The digit of a key is taken from key - base (the subtraction sits where the digit is computed, not behind the loads:
    // there it made every load wait for its own data).  Padding keys (0xFFFFFFFF) keep the top digit.
    auto digit_of = [&](uint32_t k) -> uint32_t { return ((k == 0xFFFFFFFFu ? k : k - kbase) >> shift) & DMASK; };
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const uint32_t pos = wave_base + j * 64;
        val[j] = iota ? pos : (pos < count ? vals_in[pos] : 0u);
    }
    uint32_t aux[CARRY ? KPT : 1];
    if (CARRY) {
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const uint32_t pos = wave_base + j * 64;
            aux[j] = pos < count ? aux_in[pos] : 0u;
        }
    }
    uint32_t my_tile_off = 0u;  // issued early: needed only after the ranking
    if ((uint32_t)tid <= DMASK) my_tile_off = tile_off[(size_t)tid * tiles_cap + t];
#pragma unroll
    for (int w = 0; w < WAVES; ++w) s_wave_hist[w][tid] = 0u;
    __syncthreads();

    // ---- rank inside the wave.  Phase 1, registers only: 8 ballots per key give the set of lanes holding the same
    // digit -> number of such lanes below this one, their count, and the lowest of them (the leader).  Phase 2: the
    // leaders add the counts to the wave's LDS digit counters, ALL keys' atomics issued back to back (one lane per
    // distinct digit, so no two lanes of an instruction hit the same word; LDS executes a wave's instructions in
    // order, so key j+1 sees key j: deterministic and stable).  Phase 3: the old counter value travels from the
    // leader to its group.  Two LDS round trips per tile instead of one dependent read-modify-write per key.
    uint32_t info[KPT];  // below | leader << 8 | count << 16 (count only on the leader lane, else 0)
    // The match works on the two 32-bit halves of the ballot directly: per digit bit one compare (the ballot, an
    // SGPR pair that is consumed at once), one sign-extended bit extract and two xnor/and pairs.  The scheduling
    // barrier after each key keeps the compiler from hoisting all 8 x KPT ballots first: 128 live SGPRs do not
    // exist, and the resulting v_writelane/v_readlane spill code was a fifth of this kernel's VALU time (profiles/).
    const uint32_t lt_lo = lane < 32 ? ((1u << lane) - 1u) : 0xFFFFFFFFu;
    const uint32_t lt_hi = lane < 32 ? 0u : ((1u << (lane - 32)) - 1u);
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const uint32_t d = digit_of(key[j]);
        uint32_t mlo = 0xFFFFFFFFu, mhi = 0xFFFFFFFFu;
#pragma unroll
        for (int bit = 0; bit < BITS; ++bit) {
            const uint32_t B = (uint32_t)(-(int32_t)((d >> bit) & 1u));  // all ones if the bit is set
            const unsigned long long bal = __ballot(B != 0u);
            mlo &= ~((uint32_t)bal ^ B);
            mhi &= ~((uint32_t)(bal >> 32) ^ B);
        }
        const uint32_t below = (uint32_t)__popc(mlo & lt_lo) + (uint32_t)__popc(mhi & lt_hi);
        const uint32_t leader = mlo ? (uint32_t)(__ffs((int)mlo) - 1) : 32u + (uint32_t)(__ffs((int)mhi) - 1);
        const uint32_t cnt = (below == 0u) ? (uint32_t)(__popc(mlo) + __popc(mhi)) : 0u;  // below == 0 <=> leader
        info[j] = below | (leader << 8) | (cnt << 16);
        __builtin_amdgcn_sched_barrier(0);
    }
    uint32_t prev[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const uint32_t d = digit_of(key[j]);
        prev[j] = 0u;
        if (info[j] >> 16) prev[j] = atomicAdd(&s_wave_hist[wave][d], info[j] >> 16);
    }
    uint32_t rank[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) rank[j] = __shfl(prev[j], (int)((info[j] >> 8) & 63u), 64) + (info[j] & 63u);
    __syncthreads();

    // ---- per digit (thread d = digit d): prefix over waves, tile count --------------------------------
    uint32_t tile_cnt = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        const uint32_t c = s_wave_hist[w][tid];
        s_wave_hist[w][tid] = tile_cnt;
        tile_cnt += c;
    }
    {
        const uint32_t local_excl = block_exclusive_scan(tile_cnt, s_tmp, nullptr);
        s_local_excl[tid] = local_excl;
        s_global_base[tid] = digit_base + my_tile_off - local_excl;
    }
    __syncthreads();

    // ---- reorder keys AND values through LDS (one barrier), write contiguous digit runs -------------------
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const uint32_t d = digit_of(key[j]);
        const uint32_t lpos = s_local_excl[d] + s_wave_hist[wave][d] + rank[j];
        s_keys[lpos] = key[j];
        s_vals[lpos] = val[j];
        if (CARRY) s_aux[lpos] = aux[j];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const uint32_t lp = k * SORT_THREADS + tid;
        const uint32_t kk = s_keys[lp];
        const uint32_t vv = s_vals[lp];
        const uint32_t d = digit_of(kk);
        const uint32_t gpos = s_global_base[d] + lp;
        if (lp < valid) {
            vals_out[gpos] = vv;
            if (CARRY) aux_out[gpos] = s_aux[lp];
            if (!RANGES) {
                if (KEY16) reinterpret_cast<uint16_t*>(keys_out)[gpos] = (uint16_t)kk;
                else keys_out[gpos] = kk;
            } else if (kk < nranges) {
                const uint32_t prev_k = lp > 0u ? s_keys[lp - 1u] : ~kk;
                const uint32_t next_k = lp + 1u < valid ? s_keys[lp + 1u] : ~kk;
                if (prev_k != kk) atomicMax(&ranges[kk].x, 0xFFFFFFFFu - gpos);
                if (next_k != kk) atomicMax(&ranges[kk].y, gpos + 1u);
            }
        }
    }
    __syncthreads();      // LDS is reused by the next tile
    }
}

// =====================================================================================================================
// 9-bit digits for the depth sort (round 6; verdict r05 item 4).  key - base spans less than 2^27 on every frame of every
// BASELINE workload, so THREE 9-bit passes sort it: nine launches and 76 B per key moved instead of twelve
// and 100 on the frames the 8-bit form cannot shorten (half of hd1m's views, all of c5's).  A fourth pass over bits 27..31 is
// enqueued for generality and leaves at once when the range says so (depth_range_decide).  512 digit rows against 256
// threads: every thread owns the digit pair (2 tid, 2 tid + 1) wherever the 8-bit kernels have one digit per thread.  These
// are separate kernels, not another instantiation of the ones above: the tile-id sort's instantiations stay instruction
// for instruction what they were.
// =====================================================================================================================
constexpr int RADIX9 = 512;

template <int KPT>
__global__ __launch_bounds__(SORT_THREADS) void k_dsort9_tile_hist(const uint32_t* __restrict__ keys,
                                                                  const uint32_t* __restrict__ d_count, uint32_t n, int shift,
                                                                  uint32_t* __restrict__ tile_sums, uint32_t tiles_cap,
                                                                  FrameCounters* __restrict__ fold,
                                                                  const uint32_t* __restrict__ skip,
                                                                  const uint32_t* __restrict__ key_base) {
    WS_SETPRIO_SMALL();
    constexpr int TILE_N = SORT_THREADS * KPT;
    constexpr int COPIES = 4;  // (512 digits x 4 copies = 8 KB, as the 8-bit kernel's 256 x 8)
    __shared__ uint32_t sh[RADIX9 * COPIES];
    __shared__ uint32_t s_rng[2][WAVES];
    if (skip && *skip) return;
    const uint32_t kbase = key_base ? *key_base : 0u;
    const uint32_t count = device_count(d_count, n);
    const uint32_t copy = threadIdx.x & (COPIES - 1);
    const uint32_t ntiles = (count + TILE_N - 1) / TILE_N;
    for (uint32_t L = blockIdx.x; (L >> 3) < ((ntiles + 7u) >> 3); L += gridDim.x) {
        uint32_t t;
        if (!xcd_tile(L, ntiles, &t)) continue;  // block-uniform
        for (int i = threadIdx.x; i < RADIX9 * COPIES; i += SORT_THREADS) sh[i] = 0u;
        __syncthreads();
        const uint32_t base = t * TILE_N;
        uint32_t k[KPT];
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const uint32_t pos = base + j * SORT_THREADS + threadIdx.x;
            k[j] = keys[pos < count ? pos : count - 1u];
        }
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const uint32_t pos = base + j * SORT_THREADS + threadIdx.x;
            if (pos < count) atomicAdd(&sh[(((k[j] - kbase) >> shift) & (RADIX9 - 1)) * COPIES + copy], 1u);
        }
        if (fold) {  // the frame's key range, as k_sort_tile_hist leaves it (first pass only)
            uint32_t knmin = 0u, kmax = 0u;
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                const uint32_t pos = base + j * SORT_THREADS + threadIdx.x;
                if (pos < count) {
                    knmin = max(knmin, ~k[j]);
                    kmax = max(kmax, k[j]);
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                knmin = max(knmin, (uint32_t)__shfl_xor((int)knmin, o, 64));
                kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, o, 64));
            }
            if ((threadIdx.x & 63) == 0) {
                s_rng[0][threadIdx.x >> 6] = knmin;
                s_rng[1][threadIdx.x >> 6] = kmax;
            }
        }
        __syncthreads();
        if (fold && threadIdx.x == 0) {
            uint32_t knmin = 0u, kmax = 0u;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                knmin = max(knmin, s_rng[0][w]);
                kmax = max(kmax, s_rng[1][w]);
            }
            if (knmin) {
                uint32_t* ts = fold->tile_sums + (t & (TILE_SUM_SLOTS - 1)) * TILE_SUM_STRIDE;
                atomicMax(ts + 2, knmin);
                atomicMax(ts + 3, kmax);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t d = threadIdx.x + h * SORT_THREADS;
            uint32_t c = 0;
#pragma unroll
            for (int r = 0; r < COPIES; ++r) c += sh[d * COPIES + r];
            tile_sums[(size_t)d * tiles_cap + t] = c;
        }
        __syncthreads();
    }
}

// One 9-bit pass of the depth sort: (key, value, companion) triples, digits from key - *key_base (nullptr: from the key).
template <int KPT>
__global__ __launch_bounds__(SORT_THREADS) void k_dsort9_scatter(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, const uint32_t* __restrict__ aux_in, uint32_t* __restrict__ aux_out,
    const uint32_t* __restrict__ d_count, uint32_t n, int shift, int iota, const uint32_t* __restrict__ hist,
    const uint32_t* __restrict__ tile_off, uint32_t tiles_cap, const uint32_t* __restrict__ skip,
    const uint32_t* __restrict__ key_base) {
    WS_SETPRIO_SMALL();
    constexpr int TILE_N = SORT_THREADS * KPT;
    constexpr int BITS = 9;
    constexpr uint32_t DMASK = RADIX9 - 1;
    __shared__ __attribute__((aligned(8))) uint32_t s_wave_hist[WAVES][RADIX9];
    __shared__ __attribute__((aligned(8))) uint32_t s_local_excl[RADIX9];
    __shared__ __attribute__((aligned(8))) uint32_t s_global_base[RADIX9];
    __shared__ uint32_t s_keys[TILE_N];
    __shared__ uint32_t s_vals[TILE_N];
    __shared__ uint32_t s_aux[TILE_N];
    __shared__ uint32_t s_tmp[WAVES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    if (skip && *skip) return;  // (block-uniform) a pass over a constant digit moves nothing
    const uint32_t kbase = key_base ? *key_base : 0u;
    const uint32_t count = device_count(d_count, n);
    // first output position of the digit pair (2 tid, 2 tid + 1): the same for all tiles of this pass
    const uint2 hh = reinterpret_cast<const uint2*>(hist)[tid];
    const uint32_t digit_base0 = block_exclusive_scan(hh.x + hh.y, s_tmp, nullptr);
    const uint32_t digit_base1 = digit_base0 + hh.x;
    const bool has_aux = aux_in != nullptr;  // (kernel-uniform; the standalone sorter may carry no companion)
    const uint32_t ntiles = (count + TILE_N - 1) / TILE_N;
    for (uint32_t L = blockIdx.x; (L >> 3) < ((ntiles + 7u) >> 3); L += gridDim.x) {
    uint32_t t;
    if (!xcd_tile(L, ntiles, &t)) continue;  // block-uniform
    const uint32_t tile_base = t * TILE_N;
    const uint32_t valid = (count - tile_base) < (uint32_t)TILE_N ? (count - tile_base) : (uint32_t)TILE_N;

    uint32_t key[KPT];
    uint32_t val[KPT];
    const uint32_t wave_base = tile_base + wave * (64 * KPT) + lane;
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const uint32_t pos = wave_base + j * 64;
        key[j] = pos < count ? keys_in[pos] : 0xFFFFFFFFu;
    }
    auto digit_of = [&](uint32_t k) -> uint32_t { return ((k == 0xFFFFFFFFu ? k : k - kbase) >> shift) & DMASK; };
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const uint32_t pos = wave_base + j * 64;
        val[j] = iota ? pos : (pos < count ? vals_in[pos] : 0u);
    }
    uint32_t aux[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const uint32_t pos = wave_base + j * 64;
        aux[j] = (has_aux && pos < count) ? aux_in[pos] : 0u;
    }
    // issued early: needed only after the ranking
    const uint32_t my_tile_off0 = tile_off[(size_t)(2 * tid) * tiles_cap + t];
    const uint32_t my_tile_off1 = tile_off[(size_t)(2 * tid + 1) * tiles_cap + t];
#pragma unroll
    for (int w = 0; w < WAVES; ++w) reinterpret_cast<uint2*>(s_wave_hist[w])[tid] = make_uint2(0u, 0u);
    __syncthreads();

    // ---- rank inside the wave: nine ballots per key (see k_sort_scatter) ---------------------------------------------------
    uint32_t info[KPT];
    const uint32_t lt_lo = lane < 32 ? ((1u << lane) - 1u) : 0xFFFFFFFFu;
    const uint32_t lt_hi = lane < 32 ? 0u : ((1u << (lane - 32)) - 1u);
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const uint32_t d = digit_of(key[j]);
        uint32_t mlo = 0xFFFFFFFFu, mhi = 0xFFFFFFFFu;
#pragma unroll
        for (int bit = 0; bit < BITS; ++bit) {
            const uint32_t B = (uint32_t)(-(int32_t)((d >> bit) & 1u));
            const unsigned long long bal = __ballot(B != 0u);
            mlo &= ~((uint32_t)bal ^ B);
            mhi &= ~((uint32_t)(bal >> 32) ^ B);
        }
        const uint32_t below = (uint32_t)__popc(mlo & lt_lo) + (uint32_t)__popc(mhi & lt_hi);
        const uint32_t leader = mlo ? (uint32_t)(__ffs((int)mlo) - 1) : 32u + (uint32_t)(__ffs((int)mhi) - 1);
        const uint32_t cnt = (below == 0u) ? (uint32_t)(__popc(mlo) + __popc(mhi)) : 0u;
        info[j] = below | (leader << 8) | (cnt << 16);
        __builtin_amdgcn_sched_barrier(0);
    }
    uint32_t prev[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const uint32_t d = digit_of(key[j]);
        prev[j] = 0u;
        if (info[j] >> 16) prev[j] = atomicAdd(&s_wave_hist[wave][d], info[j] >> 16);
    }
    uint32_t rank[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) rank[j] = __shfl(prev[j], (int)((info[j] >> 8) & 63u), 64) + (info[j] & 63u);
    __syncthreads();

    // ---- per digit pair: prefix over waves, tile counts --------------------------------------------------------------------
    uint32_t cnt0 = 0u, cnt1 = 0u;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        uint2* wp = reinterpret_cast<uint2*>(s_wave_hist[w]) + tid;
        const uint2 c = *wp;
        *wp = make_uint2(cnt0, cnt1);
        cnt0 += c.x;
        cnt1 += c.y;
    }
    {
        const uint32_t local_excl = block_exclusive_scan(cnt0 + cnt1, s_tmp, nullptr);
        reinterpret_cast<uint2*>(s_local_excl)[tid] = make_uint2(local_excl, local_excl + cnt0);
        reinterpret_cast<uint2*>(s_global_base)[tid] =
            make_uint2(digit_base0 + my_tile_off0 - local_excl, digit_base1 + my_tile_off1 - (local_excl + cnt0));
    }
    __syncthreads();

    // ---- reorder through LDS, write contiguous digit runs ------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const uint32_t d = digit_of(key[j]);
        const uint32_t lpos = s_local_excl[d] + s_wave_hist[wave][d] + rank[j];
        s_keys[lpos] = key[j];
        s_vals[lpos] = val[j];
        s_aux[lpos] = aux[j];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const uint32_t lp = k * SORT_THREADS + tid;
        const uint32_t kk = s_keys[lp];
        const uint32_t vv = s_vals[lp];
        const uint32_t gpos = s_global_base[digit_of(kk)] + lp;
        if (lp < valid) {
            vals_out[gpos] = vv;
            if (has_aux) aux_out[gpos] = s_aux[lp];
            keys_out[gpos] = kk;
        }
    }
    __syncthreads();  // LDS is reused by the next tile
    }
}

template <int KPT, int BITS, bool KEY16>
int run_passes_scan(const SortScratch& sc, uint32_t* kin, uint32_t* vin, uint32_t* kout, uint32_t* vout, uint32_t* ain,
                    uint32_t* aout,
                    const uint32_t* d_count, uint32_t n, int begin_bit, int npass, bool implicit_iota,
                    bool first_tile_hist_ready, hipStream_t stream, uint32_t** fk, uint32_t** fv,
                    KernelMarks* km, const char* const* names, uint2* ranges, uint32_t nranges, FrameCounters* skip_top,
                    uint32_t** fk_skipped, uint32_t** fv_skipped) {
    constexpr uint32_t TILE_N = SORT_THREADS * KPT;
    const uint32_t tiles = sort_grid((n + TILE_N - 1) / TILE_N);
    for (int p = 0; p < npass; ++p) {
        const int shift = begin_bit + p * BITS;
        const int iota = (implicit_iota && p == 0) ? 1 : 0;
        // the depth sort of a frame: its first kernel decides whether the last pass has anything to do, the kernels of the last
        // pass ask (ws_internal.h depth_range_decide)
        FrameCounters* fold = (skip_top && p == 0) ? skip_top : nullptr;
        const uint32_t* skip = (skip_top && p == npass - 1) ? &skip_top->depth_skip_top : nullptr;
        const uint32_t* kbase = (skip_top && p > 0) ? &skip_top->depth_key_base : nullptr;
        if (skip) {
            if (fk_skipped) *fk_skipped = kin;
            if (fv_skipped) *fv_skipped = vin;
        }
        uint32_t* const hist_p = sc.hist + (size_t)p * sc.hist_pitch;
        if constexpr (BITS == 9) {  // the depth sort's 9-bit form: its own kernels (above)
            hipLaunchKernelGGL((k_dsort9_tile_hist<KPT>), dim3(tiles), dim3(SORT_THREADS), 0, stream, kin, d_count, n, shift, sc.tile_sums,
                               sc.tiles_cap, fold, skip, kbase);
            km_mark(km, names[0]);
            hipLaunchKernelGGL(k_sort_col_scan, dim3(RADIX9), dim3(SORT_THREADS), 0, stream, d_count, n, TILE_N, sc.tile_sums, sc.tiles_cap,
                               hist_p, skip, fold);
            km_mark(km, names[1]);
            hipLaunchKernelGGL((k_dsort9_scatter<KPT>), dim3(tiles), dim3(SORT_THREADS), 0, stream, kin, vin, kout, vout, ain, aout, d_count,
                               n, shift, iota, hist_p, sc.tile_sums, sc.tiles_cap, skip, kbase);
            km_mark(km, names[2]);
        } else {
        if (!(p == 0 && first_tile_hist_ready)) {
            hipLaunchKernelGGL((k_sort_tile_hist<KPT, KEY16>), dim3(tiles), dim3(SORT_THREADS), 0, stream, kin, d_count, n, shift,
                               (1u << BITS) - 1u, sc.tile_sums, sc.tiles_cap, fold, skip, kbase);
            km_mark(km, names[0]);
        }
        hipLaunchKernelGGL(k_sort_col_scan, dim3(1u << BITS), dim3(SORT_THREADS), 0, stream, d_count, n, TILE_N,
                           sc.tile_sums, sc.tiles_cap, hist_p, skip, fold);
        km_mark(km, names[1]);
        if (ranges && p == npass - 1)
            hipLaunchKernelGGL((k_sort_scatter<KPT, true, BITS, false, KEY16>), dim3(tiles), dim3(SORT_THREADS), 0, stream, kin, vin,
                               kout, vout, (const uint32_t*)nullptr, (uint32_t*)nullptr, d_count, n, shift, iota,
                               hist_p, sc.tile_sums, sc.tiles_cap, ranges, nranges, skip, kbase);
        else if (ain)
            hipLaunchKernelGGL((k_sort_scatter<KPT, false, BITS, true>), dim3(tiles), dim3(SORT_THREADS), 0, stream, kin,
                               vin, kout, vout, ain, aout, d_count, n, shift, iota, hist_p, sc.tile_sums,
                               sc.tiles_cap, (uint2*)nullptr, 0u, skip, kbase);
        else
            hipLaunchKernelGGL((k_sort_scatter<KPT, false, BITS, false, KEY16>), dim3(tiles), dim3(SORT_THREADS), 0, stream, kin, vin,
                               kout, vout, (const uint32_t*)nullptr, (uint32_t*)nullptr, d_count, n, shift, iota,
                               hist_p, sc.tile_sums, sc.tiles_cap, (uint2*)nullptr, 0u, skip, kbase);
        km_mark(km, names[2]);
        }
        WS_HIP(hipGetLastError());
        uint32_t* tk = kin;
        kin = kout;
        kout = tk;
        uint32_t* tv = vin;
        vin = vout;
        vout = tv;
        uint32_t* ta = ain;
        ain = aout;
        aout = ta;
    }
    *fk = kin;
    *fv = vin;
    return WS_OK;
}


#ifdef WS_EXPERIMENTAL  // measured-and-lost variants: compiled by `make experimental` only
#include "experimental/sort_variants_kernels.hip"
#endif
}  // namespace

// Grid cap of the tile-strided kernels: 8 workgroups per CU on a 256-CU part; enough to fill the chip at any
// residency these kernels reach, small enough that surplus workgroups (count << n) cost nothing measurable.
// A multiple of 8, so that a workgroup stays on its XCD's tile range across iterations (xcd_tile).
uint32_t sort_grid(uint32_t tiles) {
    const uint32_t g = ((tiles + 7u) / 8u) * 8u;
    return g < 2048u ? (g ? g : 8u) : 2048u;
}

uint32_t sort_tile_size(uint32_t n) { return n <= SORT_SMALL_MAX ? SORT_THREADS * SORT_KPT_SMALL : SORT_TILE; }

#ifdef WS_EXPERIMENTAL  // measured-and-lost variants: compiled by `make experimental` only
#include "experimental/sort_variants_host.hip"
#else  // !WS_EXPERIMENTAL: the product library does not carry the measured-and-lost sort forms
size_t fat_sort_status_words() { return 0; }
uint32_t fat_sort_grid(uint32_t, int, int) { return 0u; }
int launch_depth_sort_fat(const FatSortScratch&, uint32_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t, bool, bool, uint32_t, int,
                          hipStream_t, KernelMarks*) {
    return fail(WS_ERR_UNSUPPORTED, "the fat-tile one-sweep depth sort is only in the experimental build");
}
int launch_tile_sort_wide(const SortScratch&, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t, int, hipStream_t, KernelMarks*,
                          uint2*, uint32_t) {
    return fail(WS_ERR_UNSUPPORTED, "the single-pass tile-id sort is only in the experimental build");
}
#endif  // WS_EXPERIMENTAL

int launch_sort_pairs(const SortScratch& sc, uint32_t* keys, uint32_t* vals, const uint32_t* d_count, uint32_t n,
                      int begin_bit, int end_bit, bool implicit_iota, bool first_tile_hist_ready, hipStream_t stream,
                      uint32_t** out_keys, uint32_t** out_vals, KernelMarks* km, const char* tag, uint2* ranges,
                      uint32_t nranges, int digit_bits, bool key16, uint32_t* aux, uint32_t* aux_alt, FrameCounters* skip_top,
                      uint32_t** out_keys_skipped, uint32_t** out_vals_skipped, int kpt9) {
    // labels of the per-kernel timers: "<tag>k_sort_..." with tag = "depth:" / "tiles:"
    const bool depth = tag && tag[0] == 'd';
    static const char* const N_DEPTH[3] = {"depth:k_sort_tile_hist", "depth:k_sort_col_scan", "depth:k_sort_scatter"};
    static const char* const N_TILES[3] = {"tiles:k_sort_tile_hist", "tiles:k_sort_col_scan", "tiles:k_sort_scatter"};
    const char* const* names = depth ? N_DEPTH : N_TILES;
    if (out_keys) *out_keys = keys;
    if (out_vals) *out_vals = vals;
    if (n == 0) return WS_OK;
    if (n > sc.cap) return fail(WS_ERR_INVALID, "sort: n exceeds the scratch capacity");
    if (aux && (ranges || !aux_alt)) return fail(WS_ERR_INVALID, "sort: companion values need a scratch partner and no range recording");
    if (key16 && end_bit > 16) return fail(WS_ERR_INVALID, "sort: 16-bit keys with more than 16 key bits");
    if (key16 && aux) return fail(WS_ERR_INVALID, "sort: companion values travel with 32-bit keys only");
    if (digit_bits < 6 || digit_bits > 9) return fail(WS_ERR_INVALID, "sort: digit width must be 6, 7, 8 or (depth sort) 9 bits");
    if (digit_bits == 9 && (key16 || ranges || first_tile_hist_ready || sc.rows < 512u || sc.hist_pitch < 512u))
        return fail(WS_ERR_INVALID, "sort: 9-bit digits belong to the depth sort (32-bit keys, a scratch with 512 count rows)");
    if (begin_bit < 0 || end_bit > 32 || begin_bit >= end_bit)
        return fail(WS_ERR_INVALID, "sort: bit range must be non-empty and within [0,32]");
    if ((reinterpret_cast<uintptr_t>(keys) & 15u) != 0) return fail(WS_ERR_INVALID, "sort: keys must be 16-byte aligned");
    const int npass = (end_bit - begin_bit + digit_bits - 1) / digit_bits;
    if (npass > 4) return fail(WS_ERR_INVALID, "sort: more than four digit passes");
    if (skip_top && (npass != 4 || begin_bit != 0 || (digit_bits != RADIX_BITS && digit_bits != 9) || first_tile_hist_ready || ranges))
        return fail(WS_ERR_INVALID, "sort: the top-byte skip belongs to the four-pass depth sort");

    uint32_t* kin = keys;
    uint32_t* vin = vals;
    uint32_t* kout = sc.keys_alt;
    uint32_t* vout = sc.vals_alt;
    int rc;
    const bool big = sort_tile_size(n) == SORT_TILE;
#define WS_RUN_SCAN(KPT_, BITS_)                                                                                       \
    rc = key16 ? run_passes_scan<KPT_, BITS_, true>(sc, kin, vin, kout, vout, aux, aux_alt, d_count, n, begin_bit, npass,   \
                                                    implicit_iota, first_tile_hist_ready, stream, &kin, &vin, km, names,   \
                                                    ranges, nranges, skip_top, out_keys_skipped, out_vals_skipped)         \
               : run_passes_scan<KPT_, BITS_, false>(sc, kin, vin, kout, vout, aux, aux_alt, d_count, n, begin_bit, npass,  \
                                                     implicit_iota, first_tile_hist_ready, stream, &kin, &vin, km, names,  \
                                                     ranges, nranges, skip_top, out_keys_skipped, out_vals_skipped)
    if (digit_bits == 9) {  // (never key16: checked above)
        // (1024-pair tiles only up to SORT_SMALL_MAX keys: the count rows' pitch is sized for that, alloc_sort_scratch)
        const bool big9 = (kpt9 == SORT_KPT_SMALL && n <= SORT_SMALL_MAX) ? false : (kpt9 ? true : big);
        if (big9) rc = run_passes_scan<SORT_KPT, 9, false>(sc, kin, vin, kout, vout, aux, aux_alt, d_count, n, begin_bit, npass, implicit_iota,
                                                          false, stream, &kin, &vin, km, names, nullptr, 0u, skip_top, out_keys_skipped,
                                                          out_vals_skipped);
        else rc = run_passes_scan<SORT_KPT_SMALL, 9, false>(sc, kin, vin, kout, vout, aux, aux_alt, d_count, n, begin_bit, npass,
                                                            implicit_iota, false, stream, &kin, &vin, km, names, nullptr, 0u, skip_top,
                                                            out_keys_skipped, out_vals_skipped);
    } else if (digit_bits == 8) {
        if (big) WS_RUN_SCAN(SORT_KPT, 8); else WS_RUN_SCAN(SORT_KPT_SMALL, 8);
    } else if (digit_bits == 7) {
        if (big) WS_RUN_SCAN(SORT_KPT, 7); else WS_RUN_SCAN(SORT_KPT_SMALL, 7);
    } else {
        if (big) WS_RUN_SCAN(SORT_KPT, 6); else WS_RUN_SCAN(SORT_KPT_SMALL, 6);
    }
#undef WS_RUN_SCAN
    if (rc) return rc;
    if (out_keys) *out_keys = kin;
    if (out_vals) *out_vals = vin;
    return WS_OK;
}

}  // namespace ws
