// sort_variants_kernels.hip -- EXPERIMENTAL BUILD ONLY (make -C web-splat_amd experimental, -DWS_EXPERIMENTAL -> lib_exp/libwebsplat_hip.so).
// Measured-and-lost variant(s), kept bit-exact and tested against lib_exp (DESIGN_LOG.md); textually included by sort.hip at the
// place the code used to stand, inside namespace ws.  The product library (lib/libwebsplat_hip.so) never compiles this file.
// ---- histogram of every participating digit in one read of the keys (the fat-tile one-sweep's extra launch) ----
// LDS bins are replicated HIST_COPIES times (copy = lane & 7): depth keys and tile ids are strongly
// clustered in their upper digits, and 64 lanes hammering one LDS word serialise.
__global__ __launch_bounds__(SORT_THREADS) void k_sort_hist(const uint32_t* __restrict__ keys,
                                                           const uint32_t* __restrict__ d_count, uint32_t n,
                                                           int begin_bit, int npass, uint32_t* __restrict__ hist) {
    __shared__ uint32_t sh[4 * RADIX * HIST_COPIES];
    for (int i = threadIdx.x; i < npass * RADIX * HIST_COPIES; i += SORT_THREADS) sh[i] = 0u;
    __syncthreads();
    const uint32_t count = device_count(d_count, n);
    const uint32_t count4 = count >> 2;
    const uint32_t copy = threadIdx.x & (HIST_COPIES - 1);
    const uint4* keys4 = reinterpret_cast<const uint4*>(keys);
    for (uint32_t i = blockIdx.x * SORT_THREADS + threadIdx.x; i < count4; i += gridDim.x * SORT_THREADS) {
        const uint4 k = keys4[i];
        for (int p = 0; p < npass; ++p) {
            const int sft = begin_bit + p * RADIX_BITS;
            uint32_t* h = sh + p * RADIX * HIST_COPIES + copy;
            atomicAdd(h + ((k.x >> sft) & (RADIX - 1)) * HIST_COPIES, 1u);
            atomicAdd(h + ((k.y >> sft) & (RADIX - 1)) * HIST_COPIES, 1u);
            atomicAdd(h + ((k.z >> sft) & (RADIX - 1)) * HIST_COPIES, 1u);
            atomicAdd(h + ((k.w >> sft) & (RADIX - 1)) * HIST_COPIES, 1u);
        }
    }
    if (blockIdx.x == 0) {
        const uint32_t i = (count4 << 2) + threadIdx.x;
        if (i < count) {
            const uint32_t k = keys[i];
            for (int p = 0; p < npass; ++p)
                atomicAdd(sh + p * RADIX * HIST_COPIES + ((k >> (begin_bit + p * RADIX_BITS)) & (RADIX - 1)) * HIST_COPIES + copy, 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < npass * RADIX; i += SORT_THREADS) {
        uint32_t c = 0;
#pragma unroll
        for (int r = 0; r < HIST_COPIES; ++r) c += sh[i * HIST_COPIES + r];
        if (c) atomicAdd(&hist[i], c);
    }
}


// =====================================================================================================================
// Single-pass tile-id sort (ws_internal.h launch_tile_sort_wide): counts [tile][bins] -> column scan -> scatter.
// =====================================================================================================================
// Column scan.  One workgroup = 16 neighbouring bins (one 64-B line per tile row) x 64 runs of consecutive tiles: a wave's
// load is four full lines.  Run sums -> exclusive prefix over the 64 runs of a bin (LDS) -> the run's offsets in place.
constexpr int WIDE_SCAN_THREADS = 1024;
__global__ __launch_bounds__(WIDE_SCAN_THREADS) void k_tile_col_scan_wide(const uint32_t* __restrict__ d_count, uint32_t n,
                                                                         uint32_t tile_n, uint32_t* __restrict__ tile_sums,
                                                                         uint32_t bins, uint32_t* __restrict__ hist) {
    constexpr int RUNS = WIDE_SCAN_THREADS / 16;
    __shared__ uint32_t s_run[RUNS][17];
    const uint32_t count = device_count(d_count, n);
    const uint32_t ntiles = (count + tile_n - 1) / tile_n;
    const uint32_t dl = threadIdx.x & 15u, run = threadIdx.x >> 4;
    const uint32_t d = blockIdx.x * 16u + dl;
    const uint32_t per = (ntiles + RUNS - 1) / RUNS;  // block-uniform
    const uint32_t t0 = run * per < ntiles ? run * per : ntiles;
    const uint32_t t1 = t0 + per < ntiles ? t0 + per : ntiles;
    uint32_t* col = tile_sums + d;
    uint32_t sum = 0;
#pragma unroll 8
    for (uint32_t t = t0; t < t1; ++t) sum += col[(size_t)t * bins];
    s_run[run][dl] = sum;
    __syncthreads();
    uint32_t off = 0, total = 0;
#pragma unroll 8
    for (int r = 0; r < RUNS; ++r) {
        const uint32_t c = s_run[r][dl];
        off += (uint32_t)r < run ? c : 0u;
        total += c;
    }
#pragma unroll 8
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t c = col[(size_t)t * bins];
        col[(size_t)t * bins] = off;
        off += c;
    }
    if (run == 0) hist[d] = total;
}

// Scatter.  Thread i owns bins [8 i, 8 i + 8) for everything that is per bin and tile-independent (first output position
// of the bin = exclusive prefix of the totals; workgroup 0 also writes the ranges from it).  Per tile: the bases of its bins
// (bin base + the tile's column offset: two 16-B loads per thread), the ballot ranking of k_sort_scatter over BITS bits with
// per-wave bin counters in LDS, the prefix of those counters over the waves, and the values go straight to their final
// place: with up to 2048 bins a tile's 2048 pairs form runs of one or two, there is nothing for an LDS reorder to merge.
template <int KPT, int BITS>
__global__ __launch_bounds__(SORT_THREADS) void k_tile_scatter_wide(
    const uint16_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ vals_out,
    const uint32_t* __restrict__ d_count, uint32_t n, const uint32_t* __restrict__ hist,
    const uint32_t* __restrict__ tile_off, uint2* __restrict__ ranges, uint32_t nranges) {
    constexpr int TILE_N = SORT_THREADS * KPT;
    constexpr int BINS = 1 << BITS;
    constexpr int BPT = BINS / SORT_THREADS > 0 ? BINS / SORT_THREADS : 1;  // bins per thread (BITS >= 8), else one per thread < BINS
    static_assert(BITS >= 7 && BITS <= TILE_SORT_WIDE_MAX_BITS, "single-pass tile sort: 7..11 bits");
    // per-wave bin counters, two waves per word (a wave holds 64 * KPT <= 65535 pairs): 16 KB instead of 32 at 2048 bins
    static_assert(WAVES == 4 && 64 * KPT * WAVES <= 0xFFFF, "packed wave counters");
    __shared__ uint32_t s_wave_hist[WAVES / 2][BINS];
    __shared__ uint32_t s_base[BINS];
    __shared__ uint32_t s_tmp[WAVES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wsel = wave >> 1;
    const uint32_t wsh = (uint32_t)(wave & 1) * 16u;
    const uint32_t count = device_count(d_count, n);
    const uint32_t ntiles = (count + TILE_N - 1) / TILE_N;
    const bool owner = (uint32_t)tid * BPT < (uint32_t)BINS;  // (BITS == 7: threads 128.. own no bin)

    uint32_t gbase[BPT];
    {
        uint32_t h[BPT];
        uint32_t sum = 0;
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            h[i] = owner ? hist[tid * BPT + i] : 0u;
            sum += h[i];
        }
        uint32_t run = block_exclusive_scan(sum, s_tmp, nullptr);
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            gbase[i] = run;
            const uint32_t d = (uint32_t)(tid * BPT + i);
            if (blockIdx.x == 0 && owner && d < nranges && h[i]) ranges[d] = make_uint2(0xFFFFFFFFu - run, run + h[i]);
            run += h[i];
        }
    }
    const uint32_t lt_lo = lane < 32 ? ((1u << lane) - 1u) : 0xFFFFFFFFu;
    const uint32_t lt_hi = lane < 32 ? 0u : ((1u << (lane - 32)) - 1u);

    for (uint32_t L = blockIdx.x; (L >> 3) < ((ntiles + 7u) >> 3); L += gridDim.x) {
        uint32_t t;
        if (!xcd_tile(L, ntiles, &t)) continue;  // block-uniform
        const uint32_t tile_base = t * TILE_N;
        // ---- load (wave-striped: order inside the tile = (wave, j, lane)) ----
        uint32_t key[KPT], val[KPT];
        const uint32_t wave_base = tile_base + wave * (64 * KPT) + lane;
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const uint32_t pos = wave_base + j * 64;
            key[j] = pos < count ? (uint32_t)keys_in[pos] : (uint32_t)(BINS - 1);  // padding: ranked behind the tile's own pairs
        }
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const uint32_t pos = wave_base + j * 64;
            val[j] = pos < count ? vals_in[pos] : 0u;
        }
        if (owner) {
            const uint32_t* row = tile_off + (size_t)t * BINS + tid * BPT;
#pragma unroll
            for (int i = 0; i < BPT; ++i) s_base[tid * BPT + i] = gbase[i] + row[i];
        }
#pragma unroll
        for (int w = 0; w < WAVES / 2; ++w)
#pragma unroll
            for (int i = 0; i < (BINS + SORT_THREADS - 1) / SORT_THREADS; ++i)
                if (BINS >= SORT_THREADS || tid < BINS) s_wave_hist[w][tid + i * SORT_THREADS] = 0u;
        __syncthreads();
        // ---- rank inside the wave (k_sort_scatter's three phases, BITS ballots per pair) ----
        uint32_t info[KPT];
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const uint32_t d = key[j] & (uint32_t)(BINS - 1);
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
            const uint32_t d = key[j] & (uint32_t)(BINS - 1);
            prev[j] = 0u;
            if (info[j] >> 16) prev[j] = (atomicAdd(&s_wave_hist[wsel][d], (info[j] >> 16) << wsh) >> wsh) & 0xFFFFu;
        }
        uint32_t rank[KPT];
#pragma unroll
        for (int j = 0; j < KPT; ++j) rank[j] = __shfl(prev[j], (int)((info[j] >> 8) & 63u), 64) + (info[j] & 63u);
        __syncthreads();
        // ---- per bin: exclusive prefix of the wave counters over the waves (bins tid, tid + 256, ...: conflict-free) ----
#pragma unroll
        for (int i = 0; i < (BINS + SORT_THREADS - 1) / SORT_THREADS; ++i) {
            const int d = tid + i * SORT_THREADS;
            if (BINS >= SORT_THREADS || tid < BINS) {
                const uint32_t w01 = s_wave_hist[0][d], w23 = s_wave_hist[1][d];
                const uint32_t c0 = w01 & 0xFFFFu, c1 = w01 >> 16, c2 = w23 & 0xFFFFu;
                s_wave_hist[0][d] = c0 << 16;                              // wave 0: 0, wave 1: c0
                s_wave_hist[1][d] = (c0 + c1) | ((c0 + c1 + c2) << 16);   // wave 2, wave 3
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const uint32_t pos = wave_base + j * 64;
            const uint32_t d = key[j] & (uint32_t)(BINS - 1);
            if (pos < count) vals_out[s_base[d] + ((s_wave_hist[wsel][d] >> wsh) & 0xFFFFu) + rank[j]] = val[j];
        }
        __syncthreads();  // LDS is reused by the next tile
    }
}

// =====================================================================================================================
// Fat-tile one-sweep depth sort (ws_internal.h FatSortScratch; WS_DEPTH_SORT=onesweep | coop)
// =====================================================================================================================
// The scan path spends twelve dependent launches on a frame's depth keys, each at or near the ~3.5-us floor of a dependent
// launch: at 0.7 M keys the sort is a chain of latencies, not of bytes.  The classic one-sweep form (chained look-back; left the tree in round 4)
// removes eight of them but walks its look-back over 330 tiles four at a time.  This form keeps the one-sweep structure --
// ONE histogram of all four digits, then one launch per digit pass, every pair read once per pass -- and makes the
// cross-tile prefix short instead:
//   * FEW, FAT chunks: <= 256 workgroups of 1024 threads, each ranks one chunk of up to 1024 x KPT pairs (the chunk size
//     follows the device-side count: ceil(count / grid) rounded up to 1024, so every workgroup has work);
//   * the prefix is a SUM, not a chain: a workgroup publishes its 256 digit counts as epoch-tagged words right after the
//     ranking and then adds up the words of ALL its predecessors, 64 rows per round trip (4 threads per digit x 16 loads in
//     flight); nobody waits for anybody's prefix, only for counts that every workgroup publishes at about the same time:
//     ceil(chunks / 64) round trips, two at 128 chunks, against ~83 serial hops of the chained form;
//   * chunks are drawn from an atomic ticket: a workgroup only waits for workgroups that already run (lookback.h).
// COOP: all four passes (and the histogram) in ONE launch, separated by device-wide barriers (grid_barrier.h): the form the
// round-3 verdict asked to be measured.  Needs every workgroup resident (grid <= CUs: checked by the launcher); several such
// launches in flight could starve each other of slots, so the per-pass launches are the production form and this one a
// measured variant (profiles/r04/).
constexpr int FAT_THREADS = 1024;
constexpr int FAT_WAVES = FAT_THREADS / 64;
constexpr int FAT_PARTS = FAT_THREADS / RADIX;  // threads per digit in the predecessor sum
constexpr int FAT_WINDOW = 16;                  // status rows per thread and round trip (64 rows per round)

// exclusive scan of one value per thread of the first 256 threads; ALL 1024 threads call it (the others pass 0)
__device__ __forceinline__ uint32_t fat_scan256(uint32_t v, uint32_t* s_tmp /*[4]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
    if (wave < 4) {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) s_tmp[wave] = incl;
    }
    __syncthreads();
    uint32_t wave_off = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) wave_off += (w < wave) ? s_tmp[w] : 0u;
    __syncthreads();  // s_tmp reusable
    return wave_off + incl - v;
}

struct FatSortArgs {
    uint32_t* keys[2];       // ping-pong: pass p reads [p & 1], writes [(p + 1) & 1]
    uint32_t* vals[2];
    uint32_t* aux[2];        // companion values (CARRY)
    const uint32_t* d_count; // device-side count (clamped to n), or nullptr
    uint32_t n;
    uint32_t* hist;          // [4][256] digit totals: filled by k_sort_hist (per-pass launches) or by phase 0 (COOP); zero on entry
    uint64_t* status;        // [4][grid][256] epoch-tagged digit counts of the chunks (never re-zeroed)
    uint32_t* tickets;       // [4] chunk dispensers, zero on entry
    uint32_t* barrier;       // COOP: gb::STATE_WORDS words, zero on entry
    uint32_t* error;         // OR-ed with 8 when a spin times out
    uint32_t epoch;
    const uint32_t* d_epoch; // != nullptr: the epoch is read here (FrameCounters::epoch, written by K1: graph replays)
    int iota;                // the payload of pass 0 is the element position
};

template <int KPT, bool CARRY, bool COOP>
__global__ __launch_bounds__(FAT_THREADS) void k_dsort_fat(const FatSortArgs a, const int pass_begin, const int pass_end) {
    constexpr int CH_MAX = FAT_THREADS * KPT;
    __shared__ uint32_t s_wave_hist[FAT_WAVES][RADIX];  // per-wave digit counters, then their prefix over the waves
    __shared__ uint32_t s_local_excl[RADIX];
    __shared__ uint32_t s_global_base[RADIX];
    __shared__ uint32_t s_part[FAT_PARTS][RADIX];
    __shared__ uint32_t s_keys[CH_MAX];
    __shared__ uint32_t s_vals[CH_MAX];
    __shared__ uint32_t s_aux[CARRY ? CH_MAX : 1];
    __shared__ uint32_t s_tmp[4];
    __shared__ uint32_t s_tile;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t grid = gridDim.x;
    const uint32_t count = device_count(a.d_count, a.n);
    const uint32_t epoch = a.d_epoch ? *a.d_epoch : a.epoch;
    // chunk size: every workgroup gets work, in whole rounds of 1024 pairs (the launcher guarantees n <= grid * CH_MAX)
    uint32_t kpt = ((count + grid - 1u) / grid + FAT_THREADS - 1u) / FAT_THREADS;
    kpt = kpt < 1u ? 1u : (kpt > (uint32_t)KPT ? (uint32_t)KPT : kpt);
    const uint32_t CH = kpt * FAT_THREADS;
    const uint32_t nchunks = (count + CH - 1u) / CH;
    // exactly nchunks tickets are drawn (by the workgroups blockIdx < nchunks); the rest have no chunk: they leave
    // (per-pass launches) or only keep the barriers company (COOP)
    const bool has_chunk = blockIdx.x < nchunks;
    if (!COOP && !has_chunk) return;
    const uint32_t lt_lo = lane < 32 ? ((1u << lane) - 1u) : 0xFFFFFFFFu;
    const uint32_t lt_hi = lane < 32 ? 0u : ((1u << (lane - 32)) - 1u);
    uint32_t t = 0u;
    uint32_t nbar = 0u;

    for (int pass = pass_begin; pass < pass_end; ++pass) {
        const int shift = pass * RADIX_BITS;
        const uint32_t* keys_in = a.keys[pass & 1];
        const uint32_t* vals_in = a.vals[pass & 1];
        const uint32_t* aux_in = a.aux[pass & 1];
        uint32_t* keys_out = a.keys[(pass + 1) & 1];
        uint32_t* vals_out = a.vals[(pass + 1) & 1];
        uint32_t* aux_out = a.aux[(pass + 1) & 1];
        if (has_chunk && (!COOP || pass == pass_begin)) {  // COOP: a workgroup keeps its chunk index through the passes
            if (tid == 0) s_tile = atomicAdd(a.tickets + pass, 1u);
            __syncthreads();
            t = s_tile;
        }
        const uint32_t chunk_base = t * CH;
        const uint32_t valid = has_chunk ? ((count - chunk_base) < CH ? (count - chunk_base) : CH) : 0u;

        // ---- load (wave-striped: order inside the chunk = (wave, j, lane) = position order) ----
        uint32_t key[KPT], val[KPT], aux[CARRY ? KPT : 1];
        const uint32_t wave_base = chunk_base + (uint32_t)wave * (64u * kpt) + (uint32_t)lane;
        if (has_chunk) {
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                const uint32_t pos = wave_base + (uint32_t)j * 64u;
                key[j] = ((uint32_t)j < kpt && pos < count) ? keys_in[pos] : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                const uint32_t pos = wave_base + (uint32_t)j * 64u;
                val[j] = (a.iota && pass == 0) ? pos : (((uint32_t)j < kpt && pos < count) ? vals_in[pos] : 0u);
                if (CARRY) aux[j] = ((uint32_t)j < kpt && pos < count) ? aux_in[pos] : 0u;
            }
        }

        if (COOP && pass == pass_begin) {
            // ---- phase 0: digit totals of ALL passes from the keys in registers (the separate histogram launch of the
            // per-pass form); LDS bins replicated four times against same-digit lanes; then a device-wide barrier
            uint32_t* sh = &s_wave_hist[0][0];  // 4 digits x 256 bins x 4 copies = 4096 words
            for (int i = tid; i < FAT_WAVES * RADIX; i += FAT_THREADS) sh[i] = 0u;
            __syncthreads();
            if (has_chunk) {
#pragma unroll
                for (int j = 0; j < KPT; ++j) {
                    const uint32_t pos = wave_base + (uint32_t)j * 64u;
                    if ((uint32_t)j < kpt && pos < count) {
#pragma unroll
                        for (int p = 0; p < 4; ++p)
                            atomicAdd(&sh[(p * RADIX + ((key[j] >> (p * RADIX_BITS)) & (RADIX - 1u))) * 4 + (lane & 3)], 1u);
                    }
                }
            }
            __syncthreads();
            {
                const uint32_t c = sh[tid * 4] + sh[tid * 4 + 1] + sh[tid * 4 + 2] + sh[tid * 4 + 3];  // bin tid of [4][256]
                if (c) atomicAdd(&a.hist[tid], c);
            }
            ws::gb::sync(a.barrier, grid, ++nbar, a.error, 8u);
        }

        // first output position of every digit: the same for all chunks of the pass
        const uint32_t digit_base = fat_scan256(tid < RADIX ? a.hist[pass * RADIX + tid] : 0u, s_tmp);

        if (has_chunk) {
#pragma unroll
            for (int i = 0; i < FAT_WAVES * RADIX / FAT_THREADS; ++i) (&s_wave_hist[0][0])[tid + i * FAT_THREADS] = 0u;
        }
        __syncthreads();

        uint32_t rank[KPT];
        if (has_chunk) {
            // ---- rank inside the wave: k_sort_scatter's three phases (ballot match, leaders bump the wave's LDS counters
            // back to back, the old counter travels to the group) ----
            uint32_t info[KPT];
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                info[j] = 0u;
                if ((uint32_t)j < kpt) {  // wave-uniform
                    const uint32_t d = (key[j] >> shift) & (RADIX - 1u);
                    uint32_t mlo = 0xFFFFFFFFu, mhi = 0xFFFFFFFFu;
#pragma unroll
                    for (int bit = 0; bit < RADIX_BITS; ++bit) {
                        const uint32_t B = (uint32_t)(-(int32_t)((d >> bit) & 1u));
                        const unsigned long long bal = __ballot(B != 0u);
                        mlo &= ~((uint32_t)bal ^ B);
                        mhi &= ~((uint32_t)(bal >> 32) ^ B);
                    }
                    const uint32_t below = (uint32_t)__popc(mlo & lt_lo) + (uint32_t)__popc(mhi & lt_hi);
                    const uint32_t leader = mlo ? (uint32_t)(__ffs((int)mlo) - 1) : 32u + (uint32_t)(__ffs((int)mhi) - 1);
                    const uint32_t cnt = (below == 0u) ? (uint32_t)(__popc(mlo) + __popc(mhi)) : 0u;
                    info[j] = below | (leader << 8) | (cnt << 16);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            uint32_t prev[KPT];
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                const uint32_t d = (key[j] >> shift) & (RADIX - 1u);
                prev[j] = 0u;
                if (info[j] >> 16) prev[j] = atomicAdd(&s_wave_hist[wave][d], info[j] >> 16);
            }
#pragma unroll
            for (int j = 0; j < KPT; ++j) rank[j] = __shfl(prev[j], (int)((info[j] >> 8) & 63u), 64) + (info[j] & 63u);
        }
        __syncthreads();

        // ---- per digit (thread d < 256): prefix over the waves, the chunk's count -> published at once ----
        uint32_t tile_cnt = 0u;
        uint64_t* my_status = a.status + ((size_t)pass * grid + t) * RADIX;
        if (has_chunk && tid < RADIX) {
#pragma unroll
            for (int w = 0; w < FAT_WAVES; ++w) {
                const uint32_t c = s_wave_hist[w][tid];
                s_wave_hist[w][tid] = tile_cnt;
                tile_cnt += c;
            }
            // padding keys (0xFFFFFFFF: digit 255 in every pass, ranked behind the chunk's own pairs) are not published
            const uint32_t pub = tile_cnt - ((tid == RADIX - 1) ? (CH - valid) : 0u);
            lb::st(my_status + tid, lb::pack(epoch, lb::FLAG_AGG, pub));
        }
        const uint32_t local_excl = fat_scan256(tile_cnt, s_tmp);

        // ---- the counts of ALL predecessors, 64 rows per round trip: thread (d, q) sums rows t-1-q, t-1-q-4, ... ----
        if (has_chunk) {
            const uint32_t d = (uint32_t)tid & (RADIX - 1u), q = (uint32_t)tid >> RADIX_BITS;
            const uint64_t* col = a.status + (size_t)pass * grid * RADIX + d;
            uint32_t sum = 0u, spins = 0u;
            int64_t r = (int64_t)t - 1 - (int64_t)q;
            while (r >= 0) {
                uint64_t w[FAT_WINDOW];
#pragma unroll
                for (int i = 0; i < FAT_WINDOW; ++i) {
                    const int64_t idx = r - (int64_t)FAT_PARTS * i;
                    w[i] = idx >= 0 ? lb::ld(col + (size_t)idx * RADIX) : lb::pack(epoch, lb::FLAG_AGG, 0u);
                }
                int consumed = 0;
#pragma unroll
                for (int i = 0; i < FAT_WINDOW; ++i) {
                    if (consumed == i && lb::flag_of(w[i], epoch) != 0u) {
                        sum += lb::value_of(w[i]);
                        consumed = i + 1;
                    }
                }
                r -= (int64_t)FAT_PARTS * consumed;
                if (consumed < FAT_WINDOW && r >= 0) {
                    if (++spins > lb::SPIN_LIMIT) {
                        if (a.error) atomicOr(a.error, 8u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            s_part[q][d] = sum;
        }
        __syncthreads();
        if (has_chunk && tid < RADIX) {
            uint32_t prev_sum = 0u;
#pragma unroll
            for (int q = 0; q < FAT_PARTS; ++q) prev_sum += s_part[q][tid];
            s_local_excl[tid] = local_excl;
            s_global_base[tid] = digit_base + prev_sum - local_excl;  // + position in the LDS-ordered chunk = output address
        }
        __syncthreads();

        // ---- reorder keys, payload (and companion) through LDS, write contiguous digit runs ----
        if (has_chunk) {
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                if ((uint32_t)j < kpt) {
                    const uint32_t d = (key[j] >> shift) & (RADIX - 1u);
                    const uint32_t lpos = s_local_excl[d] + s_wave_hist[wave][d] + rank[j];
                    s_keys[lpos] = key[j];
                    s_vals[lpos] = val[j];
                    if (CARRY) s_aux[lpos] = aux[j];
                }
            }
        }
        __syncthreads();
        if (has_chunk) {
#pragma unroll
            for (int k = 0; k < KPT; ++k) {
                const uint32_t lp = (uint32_t)k * FAT_THREADS + (uint32_t)tid;
                if ((uint32_t)k < kpt && lp < valid) {
                    const uint32_t kk = s_keys[lp];
                    const uint32_t gpos = s_global_base[(kk >> shift) & (RADIX - 1u)] + lp;
                    keys_out[gpos] = kk;
                    vals_out[gpos] = s_vals[lp];
                    if (CARRY) aux_out[gpos] = s_aux[lp];
                }
            }
        }
        if (COOP && pass + 1 < pass_end) ws::gb::sync(a.barrier, grid, ++nbar, a.error, 8u);  // (also the LDS hand-over)
        else __syncthreads();
    }
}
