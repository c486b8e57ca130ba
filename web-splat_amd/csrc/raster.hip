// raster.hip -- tile binning and per-tile front-to-back compositing for gfx950.
//
// Replaces the reference's instanced-quad draw + fixed-function blending:
//   src/shaders/gaussian.wgsl:30-67 (vs_main / fs_main), src/renderer.rs:63-67 (PREMULTIPLIED_ALPHA_BLENDING),
//   src/renderer.rs:250-260 (draw_indirect over the depth-sorted instances).
// The reference lets the ROPs read-modify-write the render target once per covered pixel per splat, back to
// front.  Here every binning tile (32x32 px by default: 2x2 tiles of 16x16) gets the list of splats that touch it (in the SAME depth order, produced by a
// stable counting sort on the tile id of depth-ordered (tile, splat) entries) and one workgroup composites the
// list front-to-back out of LDS, keeping colour and transmittance in registers and writing each pixel once.
//
//   k_bin_prefix  : per depth-sorted splat: gather its tile rectangle, count tiles, exclusive prefix over the
//                   draw order in ONE pass (ticketed wave-parallel decoupled look-back), total D
//   k_bin_emit    : entry-parallel: every workgroup produces exactly EMIT_TILE (tile id, splat) entries, whatever
//                   the footprint of the splats they come from (owners by an LDS max-scan over the splat offsets)
//   (radix sort of the entries by tile id: sort.hip, ceil(log2 T / 8) passes, stable -> depth order kept inside a
//    tile; its last pass records [begin,end) of every tile in the sorted entry list instead of writing the keys)
//   k_blend       : one workgroup per binning tile, one wave per 8x8-pixel quadrant (four waves = one 16x16 tile);
//                   splats staged through LDS 256 / 512 at a time, per-wave compaction to the records that reach
//                   the wave's quadrant, early-out on T
//   k_display     : Display::render composite into an 8-bit surface
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "blend_stage.h"
#include "emit_gen.h"
#include "lookback.h"
#include "ws_internal.h"

namespace ws {

namespace {

constexpr int BIN_THREADS = 256;
#ifndef WS_BIN_IPT
#define WS_BIN_IPT 16
#endif
constexpr int BIN_IPT = WS_BIN_IPT;                // sorted splats per thread (measured: 16 -> 25.5 us on c2, 8 -> 28.2, 4 -> 35.0:
                                                   // fewer workgroups = fewer serialised tickets, more gathers in flight)

__device__ __forceinline__ float h2f(uint32_t h) { return __half2float(__ushort_as_half((unsigned short)(h & 0xFFFFu))); }

__device__ __forceinline__ uint32_t block_exclusive_scan256(uint32_t v, uint32_t* s_tmp, uint32_t* total) {
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
    for (int w = 0; w < BIN_THREADS / 64; ++w) {
        const uint32_t c = s_tmp[w];
        if (w < wave) wave_off += c;
        tot += c;
    }
    __syncthreads();
    if (total) *total = tot;
    return wave_off + incl - v;
}

// ---- k_bin_prefix ------------------------------------------------------------------------------------
// Draw position i (far -> near) -> offsets[i] = sum of tiles touched by positions < i.  The footprint words (packed tile
// rectangles, or tile counts: ws_internal.h FootprintMode) arrive IN DRAW ORDER: they ride through the depth sort with the splat index (sort.hip), so this kernel
// streams (round 1 gathered per-splat data by sorted index at random: every XCD's L2 pulled the whole array, 3.2 x the
// algorithmic traffic on the 1 M scene and 122 of this kernel's 137 us on 5 M splats).
// Also records, for every multiple m*EMIT_TILE of the entry index, the draw position whose entry range
// contains it (emit_start[m]): the emit kernel then needs no global search to find where its slice starts.
template <int BIN_IPT>
__global__ __launch_bounds__(BIN_THREADS) void k_bin_prefix(const uint32_t* __restrict__ fp_main,
                                                           const uint32_t* __restrict__ fp_skipped, const int packed,
                                                           uint32_t* __restrict__ offsets,
                                                           uint32_t* __restrict__ emit_start,
                                                           uint64_t* __restrict__ status,
                                                           FrameCounters* __restrict__ counters, uint32_t entry_cap) {
    WS_SETPRIO_SMALL();
    constexpr int BIN_ITEMS = BIN_THREADS * BIN_IPT;
    __shared__ uint32_t s_tmp[BIN_THREADS / 64];
    __shared__ uint32_t s_bid;
    __shared__ uint32_t s_base;
    const uint32_t v = counters->num_visible;
    if ((uint64_t)blockIdx.x * BIN_ITEMS >= v) return;  // surplus workgroups leave before drawing a ticket
    // (the depth sort's last pass may have had nothing to do: the draw-ordered words are then where pass 2 left them)
    const uint32_t* __restrict__ fp_sorted = (fp_skipped && counters->depth_skip_top) ? fp_skipped : fp_main;
    const uint32_t epoch = counters->epoch;  // the frame's look-back epoch, left by K1
    const bool coarse = packed && bin_shift_decide(counters) != 0u;  // the frame bins at twice the blend's tile size (ws_internal.h)
    const int tid = threadIdx.x;
    // Workgroup ids are handed out by an atomic ticket = START order: a workgroup only ever waits for workgroups that
    // already hold their slot.  (blockIdx order -- -DWS_BLOCKIDX_ORDER -- saves the ~11 ns the returning atomic costs
    // per workgroup, in series, but is NOT safe with several look-back kernels in flight: measured on MI355X with four
    // frames on four streams, spinners of one kernel held the slots the missing predecessor of another needed and
    // vice versa -- 6 ms and 38 ms per frame instead of 0.18 and 0.35, lookback.h.)
#ifdef WS_BLOCKIDX_ORDER
    if (tid == 0) s_bid = blockIdx.x;
#else
    if (tid == 0) s_bid = atomicAdd(&counters->bin_ticket, 1u);
#endif
    __syncthreads();
    const uint32_t bid = s_bid;
    const uint32_t nblocks = (v + BIN_ITEMS - 1) / BIN_ITEMS;
    const uint32_t base = bid * BIN_ITEMS;
    if (bid == 0 && tid == 0) counters->bin_shift = coarse ? 1u : 0u;  // for k_bin_emit, the blend and the host

    // STRIPED arrangement for global memory (thread t owns draw positions base + k*256 + t: every load and store of
    // a wave is one contiguous run), BLOCKED arrangement for the scan (thread t owns BIN_IPT consecutive positions); the
    // counts change arrangement through LDS.  Measured on MI355X: with blocked global accesses (lane stride 32 B for
    // the index loads, 64 B for the stores) this kernel took 176 us on 5 M splats; the look-back was not the problem.
    __shared__ uint32_t s_cnt[BIN_ITEMS + BIN_ITEMS / 32];
    auto pad = [](uint32_t i) -> uint32_t { return i + (i >> 5); };  // blocked reads: stride IPT words -> skew the banks
    uint32_t cnt[BIN_IPT];
    uint32_t r[BIN_IPT];
#pragma unroll
    for (int k = 0; k < BIN_IPT; ++k) {
        const uint32_t i = base + k * BIN_THREADS + tid;
        r[k] = fp_sorted[i < v ? i : v - 1u];
    }
#pragma unroll
    for (int k = 0; k < BIN_IPT; ++k) {
        const uint32_t i = base + k * BIN_THREADS + tid;
        cnt[k] = i < v ? (packed ? (coarse ? rect_tiles64(r[k]) : rect_tiles(r[k])) : r[k]) : 0u;
        s_cnt[pad(k * BIN_THREADS + tid)] = cnt[k];
    }
    __syncthreads();
    uint32_t loc[BIN_IPT];
    uint32_t tsum = 0;
#pragma unroll
    for (int j = 0; j < BIN_IPT; ++j) {
        loc[j] = tsum;
        tsum += s_cnt[pad(tid * BIN_IPT + j)];
    }
    uint32_t block_total;
    const uint32_t ex = block_exclusive_scan256(tsum, s_tmp, &block_total);
    if (tid == 0) lb::st(status + bid, lb::pack(epoch, bid == 0 ? lb::FLAG_INCL : lb::FLAG_AGG, block_total));
#pragma unroll
    for (int j = 0; j < BIN_IPT; ++j) s_cnt[pad(tid * BIN_IPT + j)] = ex + loc[j];  // exclusive prefix inside the block
    if (tid < 64) {
        const uint32_t excl = lb::wave_lookback(status, bid, epoch, tid, &counters->overflow, 4u);
        if (tid == 0) {
            s_base = excl;
            const uint32_t incl = excl + block_total;  // saturates at 2^30-1 inside pack(); capacity is below that
            if (bid != 0) lb::st(status + bid, lb::pack(epoch, lb::FLAG_INCL, incl));
            if (bid == nblocks - 1) {
                uint32_t d = incl;
                counters->entries_needed = incl;
                if (d > entry_cap) {
                    atomicOr(&counters->overflow, 1u);
                    d = entry_cap;
                }
                counters->num_entries = d;
            }
        }
    }
    __syncthreads();
    const uint32_t block_off = s_base;
#pragma unroll
    for (int k = 0; k < BIN_IPT; ++k) {
        const uint32_t i = base + k * BIN_THREADS + tid;
        if (i < v) {
            const uint32_t off = block_off + s_cnt[pad(k * BIN_THREADS + tid)];
            offsets[i] = off;
            if (cnt[k]) {
                // multiples of EMIT_TILE inside [off, off + cnt)
                const uint32_t m_first = (off + EMIT_TILE - 1) / EMIT_TILE;
                const uint32_t last = off + cnt[k] - 1u;
                for (uint32_t m = m_first; (uint64_t)m * EMIT_TILE <= last; ++m)
                    if ((uint64_t)m * EMIT_TILE < entry_cap) emit_start[m] = i;
            }
        }
    }
}

// ---- k_bin_emit: (tile id, splat) entries in draw order, EMIT_TILE entries per workgroup ---------------
// Entry generation lives in emit_gen.h.  Besides writing the entries this kernel counts the first sort digit of
// every slice (the tile-id sort's per-tile histogram of pass 0).  (Generating the entries inside the sort's first
// pass instead -- no entry list in memory before it is half sorted -- was measured: the 12 B per entry saved did
// not pay for generating everything twice, once to count and once to scatter.)
// WIDE (the single-pass tile-id sort, sort.hip k_tile_scatter_wide): the whole tile id is the digit -- up to
// TILE_SORT_WIDE_MAX_BINS counters per slice, written as ONE contiguous row tile_hist[slice][bin] (tile_hist_pitch = bins).
constexpr int EMIT_COPIES = 2;

template <bool PACKED, bool WIDE>
__global__ __launch_bounds__(BIN_THREADS) void k_bin_emit(const emit::Source src, uint32_t* __restrict__ entry_keys,
                                                         uint32_t* __restrict__ entry_vals,
                                                         uint32_t* __restrict__ tile_hist, uint32_t tile_hist_pitch,
                                                         uint32_t tile_hist_mask, int key16) {
    WS_SETPRIO_SMALL();
    constexpr int HIST_WORDS = WIDE ? TILE_SORT_WIDE_MAX_BINS : RADIX * EMIT_COPIES;
    __shared__ uint32_t s_off[emit::OFF_WORDS];
    __shared__ uint32_t s_own[emit::OWN_WORDS];
    __shared__ uint32_t s_hist[HIST_WORDS];
    __shared__ uint32_t s_wmax[BIN_THREADS / 64];
    // (the depth sort's last pass may have had nothing to do: the draw-ordered arrays are then where pass 2 left them.  Two
    // wave-uniform pointers, selected once -- the kernel-argument block itself stays untouched in SGPRs)
    const bool skipped = src.sorted_idx_alt && src.counters->depth_skip_top;
    const uint32_t* __restrict__ sorted_idx = skipped ? src.sorted_idx_alt : src.sorted_idx;
    const uint32_t* __restrict__ fp_sorted = skipped ? src.fp_sorted_alt : src.fp_sorted;
    const uint32_t d = src.counters->num_entries;
    const uint32_t v = src.counters->num_visible;
    const int tid = threadIdx.x;
    const uint32_t copy = (uint32_t)tid & (EMIT_COPIES - 1);
    // capped grid, workgroups stride over the slices (the host only knows the capacity, not D)
    for (uint32_t slice = blockIdx.x; (uint64_t)slice * EMIT_TILE < d; slice += gridDim.x) {
        for (int k = tid; k < (WIDE ? (int)tile_hist_pitch : RADIX * EMIT_COPIES); k += BIN_THREADS) s_hist[k] = 0u;
        const emit::Slice sl = emit::slice_setup(src, slice, d, v, s_off, s_own, s_wmax);
        // (the frame may bin at twice the blend's tile size: the same packed rectangles in units of 2 x 2 tiles)
        const bool coarse = PACKED && src.counters->bin_shift != 0u;
        const uint32_t tiles_x = coarse ? (src.tiles_x + 1u) >> 1 : src.tiles_x;
        {
            // All EPT entries of a thread at once: owners from LDS, then their index gathers in flight together, then the
            // geometry words of their Splat records.  (Entry by entry, each one waited for its dependent loads before
            // the next one's were issued: 16 serial round trips per thread -- the whole duration of this kernel.)
            uint32_t lo[emit::EPT], val[emit::EPT], rect[PACKED ? emit::EPT : 1];
            emit::Geom geom[PACKED ? 1 : emit::EPT];
#pragma unroll
            for (int j = 0; j < emit::EPT; ++j) {
                const uint32_t el = (uint32_t)tid + (uint32_t)j * BIN_THREADS;
                lo[j] = el < sl.ne ? s_own[emit::pad(el)] : 0u;
            }
#pragma unroll
            for (int j = 0; j < emit::EPT; ++j) {
                const uint32_t pos = sl.s_lo + lo[j];
                if (PACKED) rect[j] = fp_sorted[pos];
                val[j] = sorted_idx[pos];
            }
            if (!PACKED) {
#pragma unroll
                for (int j = 0; j < emit::EPT; ++j) geom[j] = emit::load_geom(src, val[j]);
            }
#pragma unroll
            for (int j = 0; j < emit::EPT; ++j) {
                const uint32_t el = (uint32_t)tid + (uint32_t)j * BIN_THREADS;
                if (el < sl.ne) {
                    const uint32_t e = sl.e0 + el;
                    const uint32_t k = e - sl.off_at(s_off, lo[j]);
                    const uint32_t key = PACKED ? emit::tile_of_rect(coarse ? rect_coarse(rect[j]) : rect[j], k, tiles_x)
                                                : emit::tile_of(src, geom[j], k);
                    if (key16) reinterpret_cast<uint16_t*>(entry_keys)[e] = (uint16_t)key;
                    else entry_keys[e] = key;
                    entry_vals[e] = val[j];
                    if (WIDE) atomicAdd(&s_hist[key], 1u);  // (key < tiles <= bins)
                    else atomicAdd(&s_hist[(key & tile_hist_mask) * EMIT_COPIES + copy], 1u);
                }
            }
        }
        if (WIDE) {  // bin counts of sort tile `slice` for the single-pass tile-id sort ([tile][bin])
            __syncthreads();
            for (uint32_t k = (uint32_t)tid; k < tile_hist_pitch; k += BIN_THREADS)
                tile_hist[(size_t)slice * tile_hist_pitch + k] = s_hist[k];
        } else if (tile_hist) {  // digit counts of sort tile `slice` for the tile-id sort's first pass ([digit][tile])
            __syncthreads();
            uint32_t c = 0;
#pragma unroll
            for (int r = 0; r < EMIT_COPIES; ++r) c += s_hist[tid * EMIT_COPIES + r];
            if ((uint32_t)tid <= tile_hist_mask) tile_hist[(size_t)tid * tile_hist_pitch + slice] = c;
        }
        __syncthreads();  // LDS is reused by the next slice
    }
}

// ---- k_blend ---------------------------------------------------------------------------------------
// One workgroup = one tile of QW x QH quadrants (8x8 pixels each, one wave per quadrant, one pixel per lane):
// 2x2 = the 16x16 tile of the north star, 4x2 / 4x4 = two / four such tiles sharing one binned list (fewer,
// longer lists: every (tile, splat) entry costs sort and gather traffic, and a 24-px splat touches 8.3 16x16
// tiles but only 4.4 32x16 ones).  Splats are staged through LDS STAGE at a time in NEAR -> FAR order.  A staged
// record is the affine map
//   screen_pos * sqrt(log2 e) = I' * pixel_local + c      (I' = sqrt(log2 e) * M^-1, c = -I' * centre_local)
// in TILE-LOCAL pixel coordinates, so a' = |.|^2 = log2(e) * dot(screen_pos, screen_pos) of gaussian.wgsl:60 costs
// four FMAs + a multiply-add per pixel and exp(-a) is a bare v_exp_f32 (2^-a').  Measured on MI355X (profiles/):
// walking all staged records with one dependent LDS read + branch each made the kernel latency-bound
// (~30 us per batch); here the staging thread also computes which quadrants the kept ELLIPSE reaches (exact
// band-by-band x-range, blend_stage.h -- not the bounding box), every wave compacts the staged records to those
// that reach its quadrant -- one LDS read + one ballot per 64 records -- and then walks only those, with the next
// record's LDS reads in flight while the current one is composited.
constexpr float LOG2E = stage::LOG2E_F;
constexpr float CUT_A2 = CUT_A * LOG2E;  // gaussian.wgsl:61 cut-off, in the exp2 domain

// blockIdx -> tiles.  64x64-pixel blocks (16 / 8 / 4 tiles) are dealt round-robin to the 8 XCDs (workgroup b runs
// on XCD b % 8 -- observed, used for locality only): a splat's tiles mostly share a block, so its 20-B record and
// the neighbouring entry lists are served by ONE L2, while every XCD gets blocks from all over the image.
// A workgroup composites `tpw` = 2^tpw_log2 tiles of its block one after the other (the block's tiles are split
// over tpb / tpw workgroups, interleaved): with tens of thousands of tiles (4K) one workgroup per tile is bound by the
// three dependent loads each workgroup starts with; here the next tile's loads are in flight while the current tile
// is composited.
struct BlendBlock {
    uint32_t bx, by;   // 64x64-px block coordinates
    uint32_t w;        // this workgroup's lane inside the block: it owns tile slots w, w + wpb, w + 2 wpb, ...
    bool valid;
};
struct BlendShape {
    uint32_t tbx_log2, tby_log2;  // tiles per block along x / y = 8 / QW, 8 / QH
    __host__ __device__ uint32_t tpb() const { return 1u << (tbx_log2 + tby_log2); }
};
__host__ __device__ inline BlendShape blend_shape(uint32_t qw, uint32_t qh) {
    BlendShape s;
    s.tbx_log2 = qw == 2u ? 2u : (qw == 4u ? 1u : 0u);
    s.tby_log2 = qh == 2u ? 2u : (qh == 4u ? 1u : 0u);
    return s;
}
__device__ __forceinline__ BlendBlock blend_block_of(uint32_t b, uint32_t tiles_x, uint32_t tiles_y, BlendShape sh,
                                                     uint32_t tpw_log2) {
    const uint32_t nbx = (tiles_x + (1u << sh.tbx_log2) - 1u) >> sh.tbx_log2;
    const uint32_t nby = (tiles_y + (1u << sh.tby_log2) - 1u) >> sh.tby_log2;
    const uint32_t wpb = sh.tpb() >> tpw_log2;
    const uint32_t xcd = b & 7u, j = b >> 3;
    const uint32_t blk = (j / wpb) * 8u + xcd;
    BlendBlock r;
    r.w = j % wpb;
    r.valid = blk < nbx * nby;
    r.bx = r.valid ? blk % nbx : 0u;
    r.by = r.valid ? blk / nbx : 0u;
    return r;
}
inline uint32_t blend_grid_blocks(uint32_t tiles_x, uint32_t tiles_y, BlendShape sh, uint32_t tpw_log2) {
    const uint32_t nbx = (tiles_x + (1u << sh.tbx_log2) - 1u) >> sh.tbx_log2;
    const uint32_t nby = (tiles_y + (1u << sh.tby_log2) - 1u) >> sh.tby_log2;
    const uint32_t nb = nbx * nby;
    return ((nb + 7u) / 8u) * 8u * (sh.tpb() >> tpw_log2);
}
// Tiles per workgroup, measured on MI355X (WS_BLEND_TPW_LOG2 overrides): one tile per workgroup gives the lowest
// frame latency up to 1080p (more tiles per workgroup serialise the per-tile barriers: 65 -> 76 us on c2 at two
// 16x16 tiles); at 4K-class tile counts (32 k tiles, mostly short lists) four tiles per workgroup are as fast alone
// and 4 % faster with several frames in flight.
inline uint32_t blend_tpw_log2(uint32_t tiles_x, uint32_t tiles_y, BlendShape sh) {
    const uint32_t want = (tiles_x * tiles_y > 16384u) ? 2u : 0u;
    const uint32_t most = sh.tbx_log2 + sh.tby_log2;
    return want < most ? want : most;
}

// Index of the binned list blend tile (tx, ty) composites: its own, the binning tile's it is a half of (split mode), or
// -- when the frame binned at twice the blend's tile size (FrameCounters::bin_shift) -- the 2 x 2 block's it belongs to.
__device__ __forceinline__ uint32_t tile_list_index(const BlendParams& p, uint32_t tx, uint32_t ty) {
    const uint32_t s = p.counters->bin_shift;
    const uint32_t btx = s ? (p.bin_tiles_x + 1u) >> 1 : p.bin_tiles_x;
    return ((ty >> p.range_row_shift) >> s) * btx + (tx >> s);
}

template <int FORMAT>
__device__ __forceinline__ void store_pixel(const BlendParams& p, uint32_t px, uint32_t py, float r, float g, float b,
                                            float al) {
    char* row = reinterpret_cast<char*>(p.out) + (size_t)py * p.pitch;
    if (FORMAT == WS_FORMAT_RGBA32_FLOAT) {
        reinterpret_cast<float4*>(row)[px] = make_float4(r, g, b, al);
    } else if (FORMAT == WS_FORMAT_RGBA16_FLOAT) {
        const uint32_t lo = (uint32_t)__half_as_ushort(__float2half_rn(r)) | ((uint32_t)__half_as_ushort(__float2half_rn(g)) << 16);
        const uint32_t hi2 = (uint32_t)__half_as_ushort(__float2half_rn(b)) | ((uint32_t)__half_as_ushort(__float2half_rn(al)) << 16);
        reinterpret_cast<uint2*>(row)[px] = make_uint2(lo, hi2);
    } else {
        auto q8 = [](float v) -> uint32_t {
            v = fminf(fmaxf(v, 0.0f), 1.0f);
            return (uint32_t)__float2int_rn(v * 255.0f);
        };
        reinterpret_cast<uint32_t*>(row)[px] = q8(r) | (q8(g) << 8) | (q8(b) << 16) | (q8(al) << 24);
    }
}

// ---- k_blend_order: the compositing workgroups in longest-list-first order --------------------------------------------------
// The hardware starts the blend's workgroups in blockIdx order, two per CU, and a frame has ~4x (1080p) more tiles than the
// chip has slots.  Tiles take 10 ... 40 us on the headline scene and 2 ... 290 us on c3, so in image order the kernel ends
// with a tail in which most slots idle while the last dense tiles finish: measured with the time-stamped build
// (profiles/r05/blend_wait_breakdown_*.json), 24 % (hd1m) and 31 % (c3) of the kernel's span.  A tile's cost grows with the
// length of its list (until the tile saturates), and the lists are known before the blend starts: ONE workgroup counting-
// sorts the tiles by list length, longest first (the classic LPT rule of makespan scheduling), and leaves for every
// blockIdx of the blend the tile it composites together with that tile's entry range -- the blend's first scalar load
// returns all three, so its start-up chain is as long as before.  The image does not depend on the order.
constexpr int ORDER_THREADS = 1024;
constexpr int ORDER_CLASSES = 2048;   // list length / 16, capped: lengths beyond 32 k share the first class
__global__ __launch_bounds__(ORDER_THREADS) void k_blend_order(const uint2* __restrict__ tile_ranges,
                                                               const FrameCounters* __restrict__ counters, uint32_t tiles_x,
                                                               uint32_t tiles_y, uint32_t bin_tiles_x, uint4* __restrict__ order,
                                                               uint32_t nblocks, int mode) {
    WS_SETPRIO_SMALL();
    __shared__ uint32_t s_cnt[ORDER_CLASSES];
    __shared__ uint32_t s_wave[ORDER_THREADS / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t c = tid; c < (uint32_t)ORDER_CLASSES; c += ORDER_THREADS) s_cnt[c] = 0u;
    __syncthreads();
    const uint32_t s = counters->bin_shift;
    const uint32_t btx = s ? (bin_tiles_x + 1u) >> 1 : bin_tiles_x;
    const uint32_t ntiles = tiles_x * tiles_y;
    // Agent-scope loads and stores throughout (they go past the per-XCD L2s, which are not coherent with one another): this
    // ONE workgroup reads ranges that the whole chip wrote with memory-side atomics over lines the frame's memset left in the
    // L2s, twice, and both reads must agree; and every XCD's blend workgroups read the table it writes.
    auto range_of = [&](uint32_t t, uint32_t& tx, uint32_t& ty) -> uint2 {
        tx = t % tiles_x;
        ty = t / tiles_x;
        const unsigned long long w = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(tile_ranges + (ty >> s) * btx + (tx >> s)),
                                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint2 r = make_uint2((uint32_t)w, (uint32_t)(w >> 32));
        r.x = r.y ? 0xFFFFFFFFu - r.x : 0u;
        return r;
    };
    auto put = [&](uint32_t pos, uint32_t code, uint2 r) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(order + pos);
        __hip_atomic_store(o, (unsigned long long)code | ((unsigned long long)r.x << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + 1, (unsigned long long)r.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto class_of = [](uint2 r) -> uint32_t {  // class 0 = the longest lists
        const uint32_t c = (r.y - r.x) >> 4;
        return (uint32_t)(ORDER_CLASSES - 1) - (c < (uint32_t)ORDER_CLASSES ? c : (uint32_t)(ORDER_CLASSES - 1));
    };
    for (uint32_t t = tid; t < ntiles; t += ORDER_THREADS) {
        uint32_t tx, ty;
        atomicAdd(&s_cnt[class_of(range_of(t, tx, ty))], 1u);
    }
    __syncthreads();
    // exclusive scan over the classes, two per thread
    const uint32_t c0 = s_cnt[2u * tid], c1 = s_cnt[2u * tid + 1u];
    uint32_t incl = c0 + c1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if (lane >= (uint32_t)d) incl += up;
    }
    if (lane == 63u) s_wave[wave] = incl;
    __syncthreads();
    uint32_t base = 0u;
    for (uint32_t w = 0; w < wave; ++w) base += s_wave[w];
    const uint32_t excl = base + incl - (c0 + c1);
    __syncthreads();
    s_cnt[2u * tid] = excl;
    s_cnt[2u * tid + 1u] = excl + c0;
    __syncthreads();
    for (uint32_t t = tid; t < ntiles; t += ORDER_THREADS) {
        uint32_t tx, ty;
        const uint2 r = range_of(t, tx, ty);
        uint32_t pos = atomicAdd(&s_cnt[class_of(r)], 1u);   // (the order inside a class is arbitrary; the image does not see it)
        // experiments (WS_BLEND_ORDER): 2 = shortest first; 3 = alternately from both ends of the sorted sequence (a long tile,
        // a short one, the next longest, the next shortest ...): every long tile still starts early, and slots retire all the time
        if (mode == 2) pos = ntiles - 1u - pos;
        else if (mode == 3) pos = pos < (ntiles + 1u) / 2u ? 2u * pos : 2u * (ntiles - 1u - pos) + 1u;
        put(pos, tx | (ty << 16), r);
    }
    for (uint32_t b = ntiles + tid; b < nblocks; b += ORDER_THREADS) put(b, 0xFFFFFFFFu, make_uint2(0u, 0u));
}

// The last kernel of a frame folds the frame's error bits (per-frame zero arena) into the renderer's sticky words, which no
// per-frame memset clears: [0] the bits, [1] the largest entry demand of an overflowed frame.  The demand is ALSO posted --
// a plain system-scope store by this one thread -- to a host-visible mailbox word (pinned, mapped memory) that the next
// prepare() reads without any device synchronisation: a renderer whose frames overflow its entry list grows the list by
// itself, whether or not its caller ever polls ws_renderer_errors.
// (the same thread, when the blend starts: "frame frame_seq of this renderer has reached its compositing pass")
__device__ __forceinline__ void post_frame_progress(const BlendParams& p) {
    if (p.progress_mailbox) {
        __hip_atomic_store(p.progress_mailbox, p.frame_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint32_t cls = p.counters->depth_span_class;  // (0: nothing to say -- the host keeps what it knew)
        if (cls) __hip_atomic_store(p.progress_mailbox + 1, cls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__device__ __forceinline__ void fold_frame_errors(const BlendParams& p, uint32_t bits) {
    atomicOr(p.sticky, bits);
    if (bits & 1u) {
        const uint32_t need = p.counters->entries_needed;
        const uint32_t before = atomicMax(p.sticky + 1, need);  // what a retry (or the next prepare) allocates
        if (p.demand_mailbox) __hip_atomic_store(p.demand_mailbox, need > before ? need : before, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// raw words of one staged entry: the 20-B Splat record (pointcloud.rs:352-358)
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
struct RawSplat {
    u32x4_t a;    // words 0..3: kept as ONE 128-bit value from the load to the decode (see blend_fetch_raw)
    uint32_t w4;
};
// This thread's entry of the batch that ends at `hi` (slot 0 = nearest): its Splat index.  The address is clamped into
// the tile's (non-empty) range, so the load never depends on a branch and can be issued batches ahead.
template <int STAGE>
__device__ __forceinline__ uint32_t blend_entry_idx(const BlendParams& p, uint2 range, uint32_t hi, int tid) {
    const uint32_t h = hi > range.x ? hi : range.x + 1u;
    const uint32_t nb = (h - range.x) < (uint32_t)STAGE ? (h - range.x) : (uint32_t)STAGE;
    const uint32_t off = (uint32_t)tid < nb ? (uint32_t)tid : nb - 1u;
    return p.entry_vals[h - 1u - off];
}
// The 20-B Splat record.  One dwordx4 + one dword, and the four words stay a single 128-bit value until the decode: as
// five scalars the compiler parked three of them in other registers right behind the load (s_waitcnt vmcnt + v_mov: the
// gather of the next batch never overlapped the walk of the current one); five separate dword loads fix that too but cost
// 2.5x the address lookups of a random gather (measured: blend +35 % on hd1m, +29 % on c3).
__device__ __forceinline__ RawSplat blend_gather(const BlendParams& p, uint32_t idx) {
    const char* sp = reinterpret_cast<const char*>(p.splats) + (size_t)idx * SPLAT_STRIDE;
    RawSplat r;
    __builtin_memcpy(&r.a, sp, 16);
    __builtin_memcpy(&r.w4, sp + 16, 4);
    return r;
}
template <int STAGE>
__device__ __forceinline__ RawSplat blend_fetch_raw(const BlendParams& p, uint2 range, uint32_t hi, int tid) {
    return blend_gather(p, blend_entry_idx<STAGE>(p, range, hi, tid));
}

// ---- gfx950 LDS-DMA staging (k_blend<..., DMA = true>) ------------------------------------------------------------------
// The gather above holds five VGPRs per thread from the load to the decode, a whole batch later; under the 64-VGPR cap
// (two 1024-thread workgroups per CU) that is what made every deeper prefetch fail.  global_load_lds_dwordx4 /
// _dword write the record straight into LDS: nothing is held while the load flies.  The destination of lane l is the
// wave-uniform base + l * size, so a wave's 64 records land in its own 64 consecutive slots of two raw planes
// (16-B part, 4-B part), and only that wave reads them back: s_waitcnt vmcnt(0) in the same wave is all the
// synchronisation they need.
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void blend_gather_lds(const BlendParams& p, uint32_t idx, void* raw4_wave, void* raw1_wave) {
    const char* sp = reinterpret_cast<const char*>(p.splats) + (size_t)idx * SPLAT_STRIDE;
    __builtin_amdgcn_global_load_lds((gptr_t)sp, (lptr_t)raw4_wave, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(sp + 16), (lptr_t)raw1_wave, 4, 0, 0);
}
// Workgroup barrier that leaves vector-memory loads in flight.  __syncthreads() carries a workgroup-scope release: with an
// LDS-DMA outstanding the compiler drains vmcnt(0) in front of it, and the prefetch stops being one.  Here: this wave's own
// LDS traffic has completed (lgkmcnt), then the bare barrier; the "memory" clobber keeps the compiler from moving LDS
// accesses across it.
__device__ __forceinline__ void wg_barrier_keep_loads() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void wait_vector_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// LDS layout of a staged batch: two planes of 16-B records with the SAME slot stride, so one byte offset
// (slot * 16, what the per-wave lists store) addresses both with immediate offsets: two ds_read_b128 per record
// (4 LDS cycles each; the LDS array serves the CU's four SIMDs, and at ~16 VALU instructions per (record, wave)
// a third read made the LDS the co-limiter).  The colour and opacity stay the f16 pairs of the Splat record
// (words 3 and 4, copied verbatim): v_fma_mix_f32 takes f16 operands, so keeping them packed costs no conversion
// and no precision.  Slot STAGE is a null record (a' far beyond the cut-off) that pads the lists to multiples of four.
struct BlendRec {
    float4 g;  // i00', i01', c0, i10'
    float4 h;  // i11', c1, (r | g << 16) f16x2, (b | alpha << 16) f16x2
};
template <int SLOTS>
__device__ __forceinline__ BlendRec blend_load_rec(const float4* s_rec, uint32_t byte_off) {
    const char* base = reinterpret_cast<const char*>(s_rec) + byte_off;
    BlendRec r;
    r.g = *reinterpret_cast<const float4*>(base);
    r.h = *reinterpret_cast<const float4*>(base + SLOTS * 16);
    return r;
}
// EXACT (round 6, verdict r05 item 6; ws_renderer_set_blend_mode(r, WS_BLEND_FAST_EXACT_CUT)): the keep / discard decision of a
// fragment whose a' lies within the rounding band of the cut-off is taken on the ORACLE's expression -- a = |M^-1 (pixel - centre)|^2
// from the un-prescaled inverse, source order, a > 2 CUTOFF discards (gaussian.wgsl:60-63; oracle/ws_oracle.c wso_render) -- re-derived
// from the Splat record, which the rare fragment (~1e-5 of them) fetches again through its list entry; exp() keeps the fast form.
// The image then equals the oracle's to the early-out bound (6.1e-5) with NO cut-off boundary pixel on the uncompressed workloads
// (profiles/r06/blend_exact_cut_ab.txt); the band test costs two VALU instructions on the taken path of every (wave, record) pair:
// blend +5 ... +7 %, frames/s -3 %.  A mode, not the default.
struct BlendExact {
    const BlendParams* p = nullptr;
    uint32_t hi = 0u;            // end of the staged batch in the entry list (slot s holds entry hi - 1 - s)
    float fx = 0.f, fy = 0.f;    // this lane's pixel centre, absolute
    float W = 0.f, H = 0.f;
};
constexpr float CUT_BAND = 4e-6f;  // relative half-width of the band (the tile-local affine form carries ~10 roundings of 6e-8)
__device__ __forceinline__ bool blend_exact_keep(const BlendExact& e, uint32_t list_off) {
#pragma clang fp contract(off)
    const uint32_t slot = (list_off >> 4) & 1023u;
    const uint32_t idx = e.p->entry_vals[e.hi - 1u - slot];
    const uint32_t* sp = reinterpret_cast<const uint32_t*>(e.p->splats + (size_t)idx * SPLAT_STRIDE);
    const uint32_t w0 = sp[0], w1 = sp[1], w2 = sp[2];
    const float v1x = h2f(w0), v1y = h2f(w0 >> 16), v2x = h2f(w1), v2y = h2f(w1 >> 16);
    const float m00 = v1x * e.W, m01 = v2x * e.W;
    const float m10 = -v1y * e.H, m11 = -v2y * e.H;
    const float det = m00 * m11 - m01 * m10;
    const float inv = 1.0f / det;
    const float cx = (h2f(w2) * 0.5f + 0.5f) * e.W;
    const float cy = (0.5f - h2f(w2 >> 16) * 0.5f) * e.H;
    const float i00 = m11 * inv, i01 = -m01 * inv, i10 = -m10 * inv, i11 = m00 * inv;
    const float dx = e.fx - cx, dy = e.fy - cy;
    const float p0 = i00 * dx + i01 * dy;
    const float p1 = i10 * dx + i11 * dy;
    const float a = p0 * p0 + p1 * p1;
    return a <= CUT_A;
}
// One (pixel, splat) pair: gaussian.wgsl:59-66 in the exp2 domain, front-to-back "over".
template <bool EXACT = false>
__device__ __forceinline__ void blend_composite(const BlendRec& r, float lx, float ly, float& T, float& cr, float& cg,
                                                float& cb, const BlendExact& ex = BlendExact{}, uint32_t list_off = 0u) {
    const float p0 = fmaf(r.g.x, lx, fmaf(r.g.y, ly, r.g.z));
    const float p1 = fmaf(r.g.w, lx, fmaf(r.h.x, ly, r.h.y));
    const float a = fmaf(p0, p0, p1 * p1);
    bool keep = a <= (EXACT ? CUT_A2 * (1.0f + CUT_BAND) : CUT_A2);
    if (EXACT) {
        if (__builtin_expect(keep && a >= CUT_A2 * (1.0f - CUT_BAND), 0)) keep = blend_exact_keep(ex, list_off);
    }
    if (keep) {
        // b = min(0.99, 2^-a' * alpha), alpha = high half of h.w.  One asm block: gfx950 needs one wait state between
        // a transcendental's result and a VALU instruction reading it, and the compiler does not look inside asm.
        float b;
        asm("v_exp_f32_e64 %0, -%1\n\ts_nop 0\n\t"
            "v_fma_mix_f32 %0, %0, %2, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n\t"
            "v_min_f32_e32 %0, 0x3f7d70a4, %0"
            : "=&v"(b)
            : "v"(a), "v"(r.h.w));
        const float wgt = b * T;
        // plain (mixed-precision) FMAs: the compiler's v_pk_fma_f32 pairing costs a v_pk_mov and issues at half rate
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "+v"(cr) : "v"(wgt), "v"(r.h.z));
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(cg) : "v"(wgt), "v"(r.h.z));
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "+v"(cb) : "v"(wgt), "v"(r.h.w));
        T -= wgt;
    }
}

#ifndef WS_BLEND_LAST_BATCH_NO_VOTE
#define WS_BLEND_LAST_BATCH_NO_VOTE 0
#endif
#ifndef WS_BLEND_COMPACT_SKIP
#define WS_BLEND_COMPACT_SKIP 1
#endif
#ifndef WS_BLEND_MINWAVES
#define WS_BLEND_MINWAVES 1
#endif
// entries staged per batch, at most (measured at 4x4: 256 -> blend +8 % on c2, +17 % on c3; 1024 does not leave LDS
// for two workgroups per CU)
#ifndef WS_BLEND_STAGE_MAX
#define WS_BLEND_STAGE_MAX 512
#endif
// ---- TIMING build of k_blend (ws_renderer_enable_blend_timing; analysis only, never on a production launch) -------------
// Wave-uniform time stamps (s_memtime: the shader clock) bracket the phases of a tile -- range load, the first batch's
// dependent gather chain, later batches' gather waits, decode, staging barrier, compaction, walk, end-of-batch vote, pixel
// store -- and every wave leaves its sums in p.debug_timing[(tile * waves + wave) * BLEND_TIMING_WORDS + ...]
// (scripts/blend_wait_breakdown.py).  The stamp waits for the wave's own scalar / LDS traffic (lgkmcnt), never for vector
// loads: the staging prefetch stays in flight across the walk exactly as in the production kernel.
template <bool ON>
__device__ __forceinline__ uint32_t blend_stamp() {
    if (!ON) return 0u;
    uint64_t t;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return (uint32_t)t;
}
template <int FORMAT, int QW, int QH, bool MULTI, bool CAPTURE, bool DMA, bool TIMING = false, bool EXACT = false>
__global__ __launch_bounds__(64 * QW * QH, (EXACT && QW * QH == 16) ? 8 : WS_BLEND_MINWAVES) void k_blend(const BlendParams p,
                                                                                                    const uint32_t tpw_log2_arg) {
    static_assert(!TIMING || (!MULTI && !DMA && !CAPTURE), "the timing build instruments the production form only");
    static_assert(!EXACT || (!CAPTURE && !DMA && !TIMING), "the exact cut-off decision belongs to the production launch");
    uint32_t tm[9] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};  // TIMING: cycles per phase, this wave (SGPRs)
    uint32_t tm_batches = 0u, tm_real0 = 0u;
    const uint32_t tm_start = blend_stamp<TIMING>();
    if (TIMING) {
        uint64_t rt;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt)::"memory");  // 100 MHz: calibrates the shader clock
        tm_real0 = (uint32_t)rt;
    }
    const uint32_t tpw_log2 = MULTI ? tpw_log2_arg : 0u;  // MULTI = several tiles per workgroup (4K-class tile counts)
    constexpr int NW = QW * QH;                  // waves = quadrants
    constexpr int NT = 64 * NW;
    constexpr int STAGE = NT < WS_BLEND_STAGE_MAX ? NT : WS_BLEND_STAGE_MAX;   // entries staged per batch
    constexpr int SLOTS = STAGE + 1;
    constexpr int TW = 8 * QW, TH = 8 * QH;
    // (a wave compacts and walks a staged batch in sub-rounds of at most LCAP records, so the lists stay small when the
    // batch is large: -DWS_BLEND_STAGE_MAX=1024 halves the per-batch barriers and still leaves LDS for two workgroups)
    constexpr int LCAP = STAGE < 512 ? STAGE : 512;

    __shared__ float4 s_rec[2 * SLOTS];
    // quadrant bits of the staged records (0 = slot unused), 16 bits each, TRANSPOSED per sub-round of LCAP slots: the
    // masks of slots lane, lane + 64, lane + 128, ... sit side by side, so a wave's compaction reads all of them with one
    // or two wide LDS loads instead of one dependent load per 64 records
    __shared__ __attribute__((aligned(16))) uint16_t s_m[STAGE];
    static_assert(LCAP % 256 == 0, "sub-round layout of the quadrant masks: four 16-bit masks per 64-bit piece");
    // per wave: byte offsets of the staged records that reach its quadrant, near -> far, padded with the null record
    __shared__ __attribute__((aligned(16))) uint32_t s_list[NW][LCAP + 16];
    __shared__ uint2 s_range[16];   // [begin, end) of this workgroup's tiles in the sorted entry list
    __shared__ uint32_t s_txy[16];  // tx | ty << 16, or 0xFFFFFFFF for a slot outside the image
    __shared__ uint32_t s_dbg_max;  // capture mode: most records any wave walked in the current batch
    // DMA: raw Splat records as the LDS-DMA leaves them, plane of the 16-B parts and plane of the 4-B parts; two buffers:
    // the tile being composited stages out of one, the first batch of the workgroup's NEXT tile lands in the other
    __shared__ __attribute__((aligned(16))) u32x4_t s_raw4[DMA ? (MULTI ? 2 : 1) * STAGE : 1];
    __shared__ uint32_t s_raw1[DMA ? (MULTI ? 2 : 1) * STAGE : 1];
    __shared__ uint32_t s_alive[2];  // DMA: "some pixel of the tile is not saturated yet", per batch parity

    // The frame's error bits (entry overflow, look-back spin time-outs) live in the per-frame zero arena; the last
    // kernel of the frame folds them into a word that survives the next frame's memset, so a batch of frames enqueued
    // back to back can be checked once at the end (ws_renderer_errors / ws_view_batch_errors).
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.sticky) {
        const uint32_t bits = p.counters->overflow;
        if (bits) fold_frame_errors(p, bits);
        post_frame_progress(p);
    }
    ws_trace_begin(p.trace);
    const BlendShape shape = blend_shape(QW, QH);
    const BlendBlock blk = blend_block_of(blockIdx.x, p.tiles_x, p.tiles_y, shape, tpw_log2);
    // Ordered launch (k_blend_order): workgroup b composites the tile the table names for b -- every b below the tile count has
    // one, whatever blend_block_of says about b's place in the image-order layout (whose grid is padded to whole 2 x 2 blocks of
    // tiles, eight blocks at a time: its invalid indices are NOT the last ones)
    const bool ordered = !MULTI && QW == 4 && QH == 4 && p.order != nullptr;
    if (!ordered && !blk.valid) return;  // block-uniform
    const uint32_t tpw = 1u << tpw_log2, wpb = shape.tpb() >> tpw_log2;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;  // (wave-uniform: lives in an SGPR)
    const int qx = wave % QW, qy = wave / QW;
    const float lx = (float)(qx * 8 + (lane & 7)) + 0.5f;  // tile-local pixel centre
    const float ly = (float)(qy * 8 + (lane >> 3)) + 0.5f;
    const uint32_t qbit = 1u << wave;
    const bool stager = NT == STAGE || tid < STAGE;  // wave-uniform
    // One tile per workgroup (!MULTI): its coordinates and range are workgroup-uniform values of blockIdx -- scalar loads,
    // no trip through LDS and no barrier in front of the first gather.  MULTI: the ranges of all my tiles up front, one
    // round trip instead of one per tile.
    uint2 range_one = make_uint2(0u, 0u);
    uint32_t code_one = 0xFFFFFFFFu;
    if (ordered) {
        // longest list first (k_blend_order): tile and range of this blockIdx in one round trip
        // (agent-scope loads: past this XCD's L2, which may still hold the line as an earlier frame's table had it)
        const unsigned long long* op = reinterpret_cast<const unsigned long long*>(p.order + blockIdx.x);
        const unsigned long long o0 = __hip_atomic_load(op, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long o1 = __hip_atomic_load(op + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        code_one = __builtin_amdgcn_readfirstlane((uint32_t)o0);
        range_one = make_uint2(__builtin_amdgcn_readfirstlane((uint32_t)(o0 >> 32)), __builtin_amdgcn_readfirstlane((uint32_t)o1));
    } else if (!MULTI) {
        const uint32_t slot = blk.w;
        const uint32_t tx = (blk.bx << shape.tbx_log2) + (slot & ((1u << shape.tbx_log2) - 1u));
        const uint32_t ty = (blk.by << shape.tby_log2) + (slot >> shape.tbx_log2);
        if (tx < p.tiles_x && ty < p.tiles_y) {
            code_one = tx | (ty << 16);
            range_one = p.tile_ranges[tile_list_index(p, tx, ty)];
            range_one.x = range_one.y ? 0xFFFFFFFFu - range_one.x : 0u;
        }
    } else if ((uint32_t)tid < tpw) {
        const uint32_t slot = blk.w + (uint32_t)tid * wpb;
        const uint32_t tx = (blk.bx << shape.tbx_log2) + (slot & ((1u << shape.tbx_log2) - 1u));
        const uint32_t ty = (blk.by << shape.tby_log2) + (slot >> shape.tbx_log2);
        uint2 range = make_uint2(0u, 0u);
        uint32_t code = 0xFFFFFFFFu;
        if (tx < p.tiles_x && ty < p.tiles_y) {
            code = tx | (ty << 16);
            range = p.tile_ranges[tile_list_index(p, tx, ty)];
            range.x = range.y ? 0xFFFFFFFFu - range.x : 0u;
        }
        s_range[tid] = range;
        s_txy[tid] = code;
    }
    if (tid == 0) {  // the null record: a' = 1e18, never inside the cut-off (visible after the first barrier: MULTI the one
                     // below, else the first staging barrier of the tile loop -- nothing reads it before)
        s_rec[STAGE] = make_float4(0.0f, 0.0f, 1.0e9f, 0.0f);
        s_rec[SLOTS + STAGE] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    if (MULTI) __syncthreads();
    uint32_t tm_last = tm_start;
    if (TIMING) {
        asm volatile("" ::"s"(range_one.x), "s"(range_one.y));  // the tile's range has arrived (scalar loads)
        const uint32_t t = blend_stamp<TIMING>();
        tm[0] = t - tm_last;
        tm_last = t;
    }
    const float W = (float)p.width, H = (float)p.height;
    uint32_t* my_list = s_list[wave];

    RawSplat raw = {{0u, 0u, 0u, 0u}, 0u};
    uint32_t rbuf = 0u;  // DMA: the raw buffer (slot offset 0 or STAGE) the current tile stages out of
    const uint32_t wslot = (uint32_t)wave * 64u;  // first raw slot of this wave (stager waves only)
    if (stager) {
        const uint2 r0 = MULTI ? s_range[0] : range_one;
        if (r0.y > r0.x) {  // (an empty tile must not touch the entry list)
            if (DMA) blend_gather_lds(p, blend_entry_idx<STAGE>(p, r0, r0.y, tid), s_raw4 + wslot, s_raw1 + wslot);
            else raw = blend_fetch_raw<STAGE>(p, r0, r0.y, tid);
        }
    }
    for (uint32_t k = 0; k < tpw; ++k) {
    const uint32_t code = MULTI ? s_txy[k] : code_one;
    const uint2 range = MULTI ? s_range[k] : range_one;
    // the first batch of the NEXT tile (entry index -> Splat record: two dependent round trips) flies while this
    // tile is composited.  (DMA: issued from the first staging step of this tile, behind its s_waitcnt vmcnt(0): every
    // older LDS-DMA into that buffer -- the unused prefetch of an earlier tile that saturated -- has landed by then.)
    RawSplat raw_next_tile = {{0u, 0u, 0u, 0u}, 0u};
    uint2 range_nt = make_uint2(0u, 0u);  // non-empty: the next tile's first batch is still to be requested (DMA)
    if (MULTI && stager && k + 1u < tpw) {
        const uint2 rn = s_range[k + 1u];
        if (rn.y > rn.x) {
            if (DMA) range_nt = rn;
            else raw_next_tile = blend_fetch_raw<STAGE>(p, rn, rn.y, tid);
        }
    }
    if (code != 0xFFFFFFFFu) {  // block-uniform
    const uint32_t tx = code & 0xFFFFu, ty = code >> 16;
    const uint32_t tile = ty * p.tiles_x + tx;
    const uint32_t px = tx * TW + qx * 8 + (lane & 7);
    const uint32_t py = ty * TH + qy * 8 + (lane >> 3);
    const bool inside = px < p.width && py < p.height;
    // Pixels outside the image start with T = 0: they accumulate nothing and count as saturated.  There is no
    // per-pixel "done" flag in the inner loop: a pixel below T_MIN keeps accumulating (its contributions are
    // below T_MIN, the reference has no cut-off at all); T only decides when a wave / the tile may stop.
    float T = inside ? 1.0f : 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f;
    const float tile_x0 = (float)(tx * TW), tile_y0 = (float)(ty * TH);

    uint32_t hi = range.y;
    // Two-deep pipeline of the staging loads: while batch b is composited, the Splat records of batch b+1 (their indices
    // arrived during batch b-1) and the entry indices of batch b+2 are in flight.
    uint32_t idx_next = 0u;
    if (stager && range.y > range.x)
        idx_next = blend_entry_idx<STAGE>(p, range, range.y - range.x > (uint32_t)STAGE ? range.y - (uint32_t)STAGE : range.x, tid);
    // capture build only (p.debug_walked): records this wave walked, and the sum over batches of the most any wave
    // walked in the batch (the lock-step cost of the per-batch barriers)
    uint32_t dbg_walked = 0u, dbg_lockstep = 0u, dbg_deepest = 0u;
    if ((CAPTURE && p.debug_walked)) {
        if (tid == 0) s_dbg_max = 0u;
        __syncthreads();
    }
    uint32_t bpar = 0u;  // DMA: parity of the batch (which s_alive word it uses)
    while (hi > range.x) {
        const uint32_t nb = (hi - range.x) < (uint32_t)STAGE ? (hi - range.x) : (uint32_t)STAGE;
        const uint32_t hi_next = hi - nb;
        if (TIMING) {
            // the batch's Splat records (and the index load behind them) have landed: first batch = the tile's dependent
            // chain index -> record, later batches = whatever the prefetch did not hide
            wait_vector_loads();
            const uint32_t t = blend_stamp<TIMING>();
            if (tm_batches) tm[2] += t - tm_last;
            else tm[1] += t - tm_last;
            tm_last = t;
            ++tm_batches;
        }
        if (stager) {
            uint32_t mask = 0u;
            uint32_t idx_nt = 0u;
            const bool fetch_nt = DMA && MULTI && range_nt.y > range_nt.x;  // (first batch of this tile only: cleared below)
            if (DMA) {
                // this wave's LDS-DMA of the batch (issued one batch, or one tile, ago) and its index loads have landed
                wait_vector_loads();
                if (fetch_nt) idx_nt = blend_entry_idx<STAGE>(p, range_nt, range_nt.y, tid);  // consumed behind the decode
                if (tid == 0) s_alive[bpar] = 0u;  // (its readers of two batches ago are long past; its writers come after the barrier)
                raw.a = s_raw4[rbuf + (uint32_t)tid];
                raw.w4 = s_raw1[rbuf + (uint32_t)tid];
            }
            if ((uint32_t)tid < nb) {
                const stage::Staged s = stage::decode<QW, QH>(raw.a.x, raw.a.y, raw.a.z, raw.a.w, raw.w4, W, H, tile_x0,
                                                              tile_y0, CUT_A2);
                mask = s.mask;
                s_rec[tid] = make_float4(s.i00, s.i01, s.c0, s.i10);
                s_rec[SLOTS + tid] = make_float4(s.i11, s.c1, __uint_as_float(raw.a.w), __uint_as_float(raw.w4));
            }
            s_m[((uint32_t)tid / LCAP) * LCAP + ((uint32_t)tid & 63u) * (LCAP / 64) + (((uint32_t)tid % LCAP) >> 6)] = (uint16_t)mask;
            if (fetch_nt) {
                blend_gather_lds(p, idx_nt, s_raw4 + (rbuf ^ (uint32_t)STAGE) + wslot, s_raw1 + (rbuf ^ (uint32_t)STAGE) + wslot);
                range_nt = make_uint2(0u, 0u);
            }
            // the next batch's gathers fly while this batch is composited; wasted only when the tile saturates first.
            // UNCONDITIONAL (the address is clamped into the tile's range): under `if (hi_next > range.x)` the compiler
            // merged the loaded words with the old ones at the join -- s_waitcnt vmcnt directly behind the loads and three
            // v_mov, i.e. the "prefetch" waited for its own data before the barrier, one exposed round trip per batch
            if (DMA) blend_gather_lds(p, idx_next, s_raw4 + rbuf + wslot, s_raw1 + rbuf + wslot);
            else raw = blend_gather(p, idx_next);
            idx_next = blend_entry_idx<STAGE>(p, range, hi_next - range.x > (uint32_t)STAGE ? hi_next - (uint32_t)STAGE : range.x, tid);
        }
        if (TIMING) {
            const uint32_t t = blend_stamp<TIMING>();
            tm[3] += t - tm_last;  // decode + LDS stores + issue of the next batch's gathers
            tm_last = t;
        }
        if (DMA) wg_barrier_keep_loads();
        else __syncthreads();
        if (TIMING) {
            const uint32_t t = blend_stamp<TIMING>();
            tm[4] += t - tm_last;  // staging barrier: waiting for the slowest stager
            tm_last = t;
        }
        const uint32_t dbg_before = dbg_walked;
        // a wave whose 64 pixels are saturated only keeps staging
        for (uint32_t sub = 0; sub < nb && __ballot(T >= T_MIN) != 0ull; sub += (uint32_t)LCAP) {
            // wave-private compaction: records whose kept ellipse reaches this quadrant, in near -> far order
            // (the masks of the sub-round's slots lane, lane + 64, ... in 64-bit pieces -- slots past nb hold 0)
            const uint2* mp = reinterpret_cast<const uint2*>(s_m + sub + (uint32_t)lane * (LCAP / 64));
            uint32_t n = 0;
            uint32_t slot16 = (sub + (uint32_t)lane) * 16u;  // byte offset of the record of round 0
            asm volatile("" : "+v"(slot16));  // (recomputed here: hoisted out of the tile loop, the eight offsets of the
                                              // unrolled rounds would hold eight registers for the whole kernel)
#pragma unroll
            for (int h = 0; h < LCAP / 256; ++h) {
#if WS_BLEND_COMPACT_SKIP
                // (a batch of at most 256 entries -- the median hd1m tile stages 197 -- has nothing in its upper slots: a scalar branch
                //  saves their mask load and four ballots)
                if (h > 0 && nb - sub <= (uint32_t)(h * 256)) break;
#endif
                const uint2 mm = mp[h];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = h * 4 + q;
                    const uint32_t word = (q & 2) ? mm.y : mm.x;
                    const bool t = (word & (qbit << ((q & 1) * 16))) != 0u;
                    const unsigned long long bal = __ballot(t);
                    const uint32_t pos = n + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    if (t) my_list[pos] = slot16 + (uint32_t)r * 1024u;
                    n += (uint32_t)__popcll(bal);
                }
            }
            if (TIMING) {
                const uint32_t t = blend_stamp<TIMING>();
                tm[5] += t - tm_last;  // compaction (mask reads, ballots, list writes)
                tm_last = t;
            }
            if (n > 0u) {
                if (lane < 4 && ((n + (uint32_t)lane) >> 2) == (n >> 2) && (n & 3u)) my_list[n + lane] = (uint32_t)STAGE * 16u;  // pad to x4
                const uint32_t n4 = (n + 3u) >> 2;
                const uint4* lp = reinterpret_cast<const uint4*>(my_list);
                // groups of four list entries; the next group's offsets and the next record are in flight while the
                // current record is composited
                uint4 o = lp[0];
                uint4 on = lp[n4 > 1u ? 1u : 0u];
                BlendRec cur = blend_load_rec<SLOTS>(s_rec, o.x);
                for (uint32_t g = 0; g < n4; ++g) {
                    const BlendExact exact = {&p, hi, tile_x0 + lx, tile_y0 + ly, W, H};  // (EXACT only; dead code otherwise)
                    const BlendRec r1 = blend_load_rec<SLOTS>(s_rec, o.y);
                    blend_composite<EXACT>(cur, lx, ly, T, cr, cg, cb, exact, o.x);
                    const BlendRec r2 = blend_load_rec<SLOTS>(s_rec, o.z);
                    blend_composite<EXACT>(r1, lx, ly, T, cr, cg, cb, exact, o.y);
                    const BlendRec r3 = blend_load_rec<SLOTS>(s_rec, o.w);
                    blend_composite<EXACT>(r2, lx, ly, T, cr, cg, cb, exact, o.z);
                    const uint32_t off3 = o.w;
                    cur = blend_load_rec<SLOTS>(s_rec, on.x);  // (re-reads a valid record after the last group)
                    blend_composite<EXACT>(r3, lx, ly, T, cr, cg, cb, exact, off3);
                    // the quadrant is saturated: nothing behind can add more than T_MIN (one compare per four pairs;
                    // on dense tiles this stops the walk well inside the staged batch)
                    if ((CAPTURE && p.debug_walked) || TIMING) dbg_walked += 4u;
                    if (CAPTURE && p.debug_consumed) {  // the deepest list position this wave has composited so far (1-based from the near end)
                        const uint32_t nul = (uint32_t)STAGE * 16u;
                        uint32_t m = o.x;               // (o.x is never the padding record)
                        m = (o.y != nul && o.y > m) ? o.y : m;
                        m = (o.z != nul && o.z > m) ? o.z : m;
                        m = (o.w != nul && o.w > m) ? o.w : m;
                        const uint32_t deep = (range.y - hi) + (m >> 4) + 1u;
                        dbg_deepest = deep > dbg_deepest ? deep : dbg_deepest;
                    }
                    if (__ballot(T >= T_MIN) == 0ull) break;
                    o = on;
                    on = lp[g + 2u < n4 ? g + 2u : n4 - 1u];
                }
            }
        }
        hi = hi_next;
        if ((CAPTURE && p.debug_walked) && lane == 0) atomicMax(&s_dbg_max, dbg_walked - dbg_before);
        if (TIMING) {
            const uint32_t t = blend_stamp<TIMING>();
            tm[6] += t - tm_last;  // walk
            tm_last = t;
        }
#if WS_BLEND_LAST_BATCH_NO_VOTE
        // The LAST batch of the list needs no vote: nobody stages behind it, so a wave stores its pixels -- and gives its wave slot
        // back -- when ITS walk ends instead of waiting for the slowest of sixteen (the median hd1m tile has one batch: all of its
        // vote wait).  (MULTI: the barrier between two tiles of a workgroup stays, below.  Capture / LDS-DMA builds keep the vote.)
        if (!DMA && !(CAPTURE && p.debug_walked) && hi == range.x) break;
#endif
        bool all_done;
        if (DMA) {  // __syncthreads_and() without the release fence that would drain the staging DMA
            const bool wave_alive = __ballot(T >= T_MIN) != 0ull;  // (evaluated by ALL lanes, not behind `lane == 0 &&`)
            if (lane == 0 && wave_alive) s_alive[bpar] = 1u;
            wg_barrier_keep_loads();
            all_done = s_alive[bpar] == 0u;
            bpar ^= 1u;
        } else {
            all_done = __syncthreads_and(T < T_MIN ? 1 : 0);
        }
        if ((CAPTURE && p.debug_walked)) {
            dbg_lockstep += s_dbg_max;
            __syncthreads();
            if (tid == 0) s_dbg_max = 0u;
            __syncthreads();
        }
        if (TIMING) {
            const uint32_t t = blend_stamp<TIMING>();
            tm[7] += t - tm_last;  // end-of-batch vote: waiting for the wave with the longest walk
            tm_last = t;
        }
        if (all_done) break;
    }
    // capture: how deep into its list the tile was read -- the deepest entry any wave composited (the walk of a wave stops inside
    // a staged batch; the staging itself runs in whole batches: range.y - hi would say 512 for most tiles)
    if ((CAPTURE && p.debug_consumed) && lane == 0 && dbg_deepest) atomicMax(&p.debug_consumed[tile], dbg_deepest);
    if ((CAPTURE && p.debug_walked) && lane == 0) {
        p.debug_walked[(size_t)tile * 17u + wave] = dbg_walked;
        if (wave == 0) p.debug_walked[(size_t)tile * 17u + 16u] = dbg_lockstep;
    }

    {
        // (the pixel coordinates are rebuilt from lx / ly instead of staying live across the walk: two registers)
        const uint32_t sx = tx * TW + (uint32_t)lx, sy = ty * TH + (uint32_t)ly;
        if (sx < p.width && sy < p.height) {
            // begin_render_pass(clear = background) then "over": dst = src + dst * (1 - src.a), all four channels
            store_pixel<FORMAT>(p, sx, sy, cr + p.background[0] * T, cg + p.background[1] * T, cb + p.background[2] * T,
                                (1.0f - T) + p.background[3] * T);
        }
    }
    if (TIMING && p.debug_timing) {
        const uint32_t t = blend_stamp<TIMING>();
        tm[8] = t - tm_last;
        uint64_t rt;
        uint32_t xcc, hw;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt)::"memory");
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        if (lane < BLEND_TIMING_WORDS) {
            uint32_t v = 0u;
#pragma unroll
            for (int i = 0; i < 9; ++i) v = lane == i ? tm[i] : v;
            v = lane == 9 ? tm_batches : v;
            v = lane == 10 ? dbg_walked : v;
            v = lane == 11 ? tm_start : v;
            v = lane == 12 ? t : v;
            v = lane == 13 ? tm_real0 : v;
            v = lane == 14 ? (uint32_t)rt : v;
            v = lane == 15 ? ((xcc & 0xFu) << 28) | (hw & 0x0FFFFFFFu) : v;
            p.debug_timing[((size_t)tile * NW + wave) * BLEND_TIMING_WORDS + lane] = v;
        }
    }
    }  // tile inside the image
    if (!MULTI) break;
    if (DMA) {
        if (stager && range_nt.y > range_nt.x) {  // this tile had no batch to issue the next tile's first batch from
            wait_vector_loads();
            blend_gather_lds(p, blend_entry_idx<STAGE>(p, range_nt, range_nt.y, tid), s_raw4 + (rbuf ^ (uint32_t)STAGE) + wslot,
                             s_raw1 + (rbuf ^ (uint32_t)STAGE) + wslot);
        }
        rbuf ^= (uint32_t)STAGE;
        wg_barrier_keep_loads();  // the staging buffers are reused by the next tile
    } else {
        raw = raw_next_tile;
        __syncthreads();  // the staging buffers are reused by the next tile
    }
    }  // tiles of this workgroup
    if (DMA) wait_vector_loads();  // a prefetch the tile did not consume must have landed before the workgroup's LDS is released
    ws_trace_end(p.trace);
}

#ifdef WS_EXPERIMENTAL  // measured-and-lost variants: compiled by `make experimental` only
#include "experimental/blend_async.hip"
#endif
// ---- k_blend_q: one WAVE per 8x8 quadrant, no LDS, no barriers -------------------------------------------
// Measured on MI355X (profiles/): the 256-thread kernel above is bound by the serial latency of one tile (two
// barriers per 256-splat batch, three dependent LDS reads per splat), not by VALU (26 % busy) or LDS bandwidth.
// Here every 8x8 quadrant is an independent 64-lane workgroup: lane l decodes entry l of the current 64-entry
// chunk into registers, a ballot keeps only the splats whose padded bounding box touches the quadrant, and the
// survivors are broadcast one at a time with v_readlane into SGPRs, which VALU instructions take as operands
// directly.  No __syncthreads, early-out per quadrant, the next chunk's gathers are in flight while the current
// chunk is composited.  The four quadrants of a tile are mapped to the same XCD (same L2) and dispatched
// close together, so the entry list and the Splat records are fetched from HBM once.
struct StagedSplat {
    float cx, cy, i00, i01, i10, i11, alpha, r, g, b;
    bool touch;
};

__device__ __forceinline__ StagedSplat decode_splat(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4,
                                                    float W, float H, float qx_lo, float qy_lo, bool valid) {
#pragma clang fp contract(off)  // (k_blend_strict: the oracle's setup_splat, operation by operation)
    StagedSplat s;
    const float v1x = h2f(w0), v1y = h2f(w0 >> 16), v2x = h2f(w1), v2y = h2f(w1 >> 16);
    const float m00 = v1x * W, m01 = v2x * W;
    const float m10 = -v1y * H, m11 = -v2y * H;
    const float det = m00 * m11 - m01 * m10;
    const float inv = 1.0f / det;
    s.cx = (h2f(w2) * 0.5f + 0.5f) * W;
    s.cy = (0.5f - h2f(w2 >> 16) * 0.5f) * H;
    s.i00 = m11 * inv;
    s.i01 = -m01 * inv;
    s.i10 = -m10 * inv;
    s.i11 = m00 * inv;
    s.alpha = h2f(w4 >> 16);
    s.r = h2f(w3);
    s.g = h2f(w3 >> 16);
    s.b = h2f(w4);
    const float rad = 2.1697873f * 1.00001f;
    const float exx = rad * sqrtf(m00 * m00 + m01 * m01) + 1e-3f;
    const float eyy = rad * sqrtf(m10 * m10 + m11 * m11) + 1e-3f;
    // pixel centres of the quadrant span [q_lo, q_lo + 7]
    s.touch = valid && (s.cx + exx >= qx_lo) && (s.cx - exx <= qx_lo + 7.0f) && (s.cy + eyy >= qy_lo) &&
              (s.cy - eyy <= qy_lo + 7.0f);
    return s;
}

__device__ __forceinline__ float bcast(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

#ifdef WS_EXPERIMENTAL  // measured-and-lost variants: compiled by `make experimental` only
#include "experimental/blend_q.hip"
#endif

// ---- k_blend_strict: the reference's blend, literally -------------------------------------------------------------
// The render pass of the reference clears the target to the background and lets the fixed-function blender apply
// PREMULTIPLIED_ALPHA_BLENDING (src/renderer.rs:63-67) once per splat, BACK TO FRONT, on a target of the pass's own
// precision: Rgba16Float in bin/render.rs:154, Rgba8Unorm in bin/measure.rs:184, i.e. the destination is rounded to
// f16 / unorm8 after EVERY splat.  k_blend rounds once, at the store (and stops early): closer to the exact integral,
// but not what the reference's binaries write.  This variant (ws_renderer_set_blend_mode(.., WS_BLEND_TARGET_PRECISION);
// the default of ws_render_views) walks each tile's list far -> near with no early-out and rounds the destination after
// every splat the way the oracle's target modes do: dst = q(src + dst * (1 - src.a)), q = f16 RNE / unorm8 RNE / identity.
// One wave per 8x8 quadrant (the shape of k_blend_q): lane = pixel, records broadcast with v_readlane; the quadrant test
// is the padded bounding box -- the per-pixel test a <= 2*CUTOFF decides, exactly as gaussian.wgsl:59-66.
// Throughput is not the point of this kernel (every entry of every list is walked).
template <int FORMAT>
__device__ __forceinline__ float quantize_target(float v) {
    if (FORMAT == WS_FORMAT_RGBA16_FLOAT) return __half2float(__float2half_rn(v));
    if (FORMAT == WS_FORMAT_RGBA8_UNORM) return rintf(fminf(fmaxf(v, 0.0f), 1.0f) * 255.0f) / 255.0f;
    return v;
}

template <int FORMAT>
__global__ __launch_bounds__(64) void k_blend_strict(const BlendParams p) {
#pragma clang fp contract(off)  // the oracle's arithmetic: separate multiplies and adds
    const uint32_t b = blockIdx.x;
    if (b == 0 && threadIdx.x == 0 && p.sticky) {  // as in k_blend
        const uint32_t bits = p.counters->overflow;
        if (bits) fold_frame_errors(p, bits);
        post_frame_progress(p);
    }
    const uint32_t xcd = b & 7u, j = b >> 3;
    const uint32_t nq = p.qw * p.qh;
    const uint32_t q = j % nq;
    const uint32_t tile = (j / nq) * 8u + xcd;
    if (tile >= p.tiles_x * p.tiles_y) return;
    const uint32_t tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int lane = threadIdx.x;
    const uint32_t qx0 = tx * p.qw * 8u + (q % p.qw) * 8u, qy0 = ty * p.qh * 8u + (q / p.qw) * 8u;
    const uint32_t px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = px < p.width && py < p.height;
    const float fx = (float)px + 0.5f, fy = (float)py + 0.5f;
    const float qx_lo = (float)qx0 + 0.5f, qy_lo = (float)qy0 + 0.5f;
    const float W = (float)p.width, H = (float)p.height;
    float d0 = quantize_target<FORMAT>(p.background[0]), d1 = quantize_target<FORMAT>(p.background[1]),
          d2 = quantize_target<FORMAT>(p.background[2]), d3 = quantize_target<FORMAT>(p.background[3]);
    uint2 range = p.tile_ranges[tile_list_index(p, tx, ty)];
    range.x = range.y ? 0xFFFFFFFFu - range.x : 0u;
    for (uint32_t lo = range.x; lo < range.y; lo += 64u) {  // far -> near: ascending position in the tile's list
        const uint32_t e = lo + (uint32_t)lane;
        const bool valid = e < range.y;
        const uint32_t idx = p.entry_vals[valid ? e : range.y - 1u];
        const uint32_t* sp = reinterpret_cast<const uint32_t*>(p.splats + (size_t)idx * SPLAT_STRIDE);
        const StagedSplat s = decode_splat(sp[0], sp[1], sp[2], sp[3], sp[4], W, H, qx_lo, qy_lo, valid);
        unsigned long long rel = __ballot(s.touch);
        while (rel) {
            const int k = __ffsll((long long)rel) - 1;
            rel &= rel - 1ull;
            // gaussian.wgsl:59-66 with the oracle's operation order (oracle/ws_oracle.c wso_render)
            const float dx = fx - bcast(s.cx, k), dy = fy - bcast(s.cy, k);
            const float p0 = bcast(s.i00, k) * dx + bcast(s.i01, k) * dy;
            const float p1 = bcast(s.i10, k) * dx + bcast(s.i11, k) * dy;
            const float a = p0 * p0 + p1 * p1;
            if (a <= CUT_A) {
                const float bb = fminf(0.99f, expf(-a) * bcast(s.alpha, k));
                const float om = 1.0f - bb;
                d0 = quantize_target<FORMAT>(bcast(s.r, k) * bb + d0 * om);
                d1 = quantize_target<FORMAT>(bcast(s.g, k) * bb + d1 * om);
                d2 = quantize_target<FORMAT>(bcast(s.b, k) * bb + d2 * om);
                d3 = quantize_target<FORMAT>(1.0f * bb + d3 * om);
            }
        }
    }
    if (inside) store_pixel<FORMAT>(p, px, py, d0, d1, d2, d3);
}

}  // namespace

// ---- k_display: Display::render (renderer.rs:548-582) + display.wgsl:37-55 -------------------------------
// A full-screen quad samples the splat image at texel centres (identity: source and target have the same size)
// and PREMULTIPLIED_ALPHA_BLENDING puts it over the surface cleared to `background`:
//   dst = src + background * (1 - src.a), then the 8-bit unorm store of the surface.  Pure streaming: 4..16 B in,
// 4 B out per pixel; one thread per pixel, rows are contiguous.
namespace {
struct DisplayParams {
    const void* src;
    void* dst;
    size_t src_pitch, dst_pitch;
    uint32_t w, h;
    float bg[4];
    int src_format, dst_format;
};
__global__ __launch_bounds__(256) void k_display(const DisplayParams p) {
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.w || y >= p.h) return;
    const char* row = reinterpret_cast<const char*>(p.src) + (size_t)y * p.src_pitch;
    float r, g, b, a;
    if (p.src_format == WS_FORMAT_RGBA32_FLOAT) {
        const float4 v = reinterpret_cast<const float4*>(row)[x];
        r = v.x; g = v.y; b = v.z; a = v.w;
    } else if (p.src_format == WS_FORMAT_RGBA16_FLOAT) {
        const uint2 v = reinterpret_cast<const uint2*>(row)[x];
        r = h2f(v.x); g = h2f(v.x >> 16); b = h2f(v.y); a = h2f(v.y >> 16);
    } else {
        const uint32_t v = reinterpret_cast<const uint32_t*>(row)[x];
        r = (float)(v & 255u) / 255.0f; g = (float)((v >> 8) & 255u) / 255.0f;
        b = (float)((v >> 16) & 255u) / 255.0f; a = (float)(v >> 24) / 255.0f;
    }
    const float k = 1.0f - a;
    const float o0 = r + p.bg[0] * k, o1 = g + p.bg[1] * k, o2 = b + p.bg[2] * k, o3 = a + p.bg[3] * k;
    auto q8 = [](float v) -> uint32_t {
        v = fminf(fmaxf(v, 0.0f), 1.0f);
        return (uint32_t)__float2int_rn(v * 255.0f);
    };
    const uint32_t out = p.dst_format == WS_SURFACE_BGRA8_UNORM ? (q8(o2) | (q8(o1) << 8) | (q8(o0) << 16) | (q8(o3) << 24))
                                                                : (q8(o0) | (q8(o1) << 8) | (q8(o2) << 16) | (q8(o3) << 24));
    reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(p.dst) + (size_t)y * p.dst_pitch)[x] = out;
}
}  // namespace

int launch_display(const void* src, int src_format, size_t src_pitch, uint32_t w, uint32_t h, const float bg[4],
                   int dst_format, void* dst, size_t dst_pitch, hipStream_t stream) {
    DisplayParams p;
    p.src = src;
    p.dst = dst;
    p.src_pitch = src_pitch;
    p.dst_pitch = dst_pitch;
    p.w = w;
    p.h = h;
    for (int i = 0; i < 4; ++i) p.bg[i] = bg[i];
    p.src_format = src_format;
    p.dst_format = dst_format;
    hipLaunchKernelGGL(k_display, dim3((w + 63) / 64, (h + 3) / 4), dim3(256), 0, stream, p);
    WS_HIP(hipGetLastError());
    return WS_OK;
}

// An empty launch: with per-kernel timers on, its event interval is the dispatch latency every dependent launch
// of the frame carries (the kernel times rocprofv3 reports do not include it).
namespace {
__global__ void k_empty() {}
}  // namespace
int launch_empty(hipStream_t stream) {
    hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, stream);
    WS_HIP(hipGetLastError());
    return WS_OK;
}

uint32_t bin_prefix_blocks(uint32_t max_points) {
    const uint32_t items = BIN_THREADS * BIN_IPT;
    return (max_points + items - 1) / items;
}

int launch_bin_prefix(const BinBuffers& b, hipStream_t stream) {
    const uint32_t blocks = bin_prefix_blocks(b.max_points);
    if (blocks == 0) return WS_OK;
    hipLaunchKernelGGL(k_bin_prefix<BIN_IPT>, dim3(blocks), dim3(BIN_THREADS), 0, stream, b.fp_sorted, b.fp_sorted_alt,
                       b.footprint_mode == FP_RECT_PACKED ? 1 : 0, b.offsets,
                       b.emit_start, b.block_status, b.counters, b.entry_cap);
    WS_HIP(hipGetLastError());
    return WS_OK;
}

int launch_bin_emit(const BinBuffers& b, hipStream_t stream) {
    uint32_t blocks = (b.entry_cap + EMIT_TILE - 1) / EMIT_TILE;
    if (blocks == 0) return WS_OK;
    if (blocks > 2048u) blocks = 2048u;  // slices are strided over; surplus workgroups are not free
    emit::Source src;
    src.sorted_idx = b.sorted_idx;
    src.fp_sorted = b.fp_sorted;
    src.sorted_idx_alt = b.sorted_idx_alt;
    src.fp_sorted_alt = b.fp_sorted_alt;
    src.splats = b.splats;
    src.ellipse = b.footprint_mode == FP_ELLIPSE;
    src.offsets = b.offsets;
    src.emit_start = b.emit_start;
    src.counters = b.counters;
    src.tiles_x = b.tiles_x;
    src.vw = b.vw;
    src.vh = b.vh;
    src.tile_w_log2 = b.tile_w_log2;
    src.tile_h_log2 = b.tile_h_log2;
    if (b.tile_hist_wide && (!b.tile_hist || !b.key16 || b.tile_hist_pitch > (uint32_t)TILE_SORT_WIDE_MAX_BINS))
        return fail(WS_ERR_INVALID, "bin emit: the single-pass tile sort needs its count rows, 16-bit keys and at most 2048 bins");
#define WS_EMIT(PACKED_, WIDE_)                                                                                        \
    hipLaunchKernelGGL((k_bin_emit<PACKED_, WIDE_>), dim3(blocks), dim3(BIN_THREADS), 0, stream, src, b.entry_keys,   \
                       b.entry_vals, b.tile_hist, b.tile_hist_pitch, b.tile_hist_mask, b.key16)
    if (b.footprint_mode == FP_RECT_PACKED) {
        if (b.tile_hist_wide) WS_EMIT(true, true); else WS_EMIT(true, false);
    } else {
        if (b.tile_hist_wide) WS_EMIT(false, true); else WS_EMIT(false, false);
    }
#undef WS_EMIT
    WS_HIP(hipGetLastError());
    return WS_OK;
}

uint32_t blend_order_blocks(uint32_t tiles_x, uint32_t tiles_y) { return blend_grid_blocks(tiles_x, tiles_y, blend_shape(4, 4), 0u); }

int launch_blend_order(const uint2* tile_ranges, const FrameCounters* counters, uint32_t tiles_x, uint32_t tiles_y, uint4* order,
                       int mode, hipStream_t stream) {
    if (tiles_x * tiles_y == 0u) return WS_OK;
    hipLaunchKernelGGL(k_blend_order, dim3(1), dim3(ORDER_THREADS), 0, stream, tile_ranges, counters, tiles_x, tiles_y, tiles_x, order,
                       blend_order_blocks(tiles_x, tiles_y), mode);
    WS_HIP(hipGetLastError());
    return WS_OK;
}

template <int QW, int QH>
static int launch_blend_shape(const BlendParams& p, hipStream_t stream) {
    const BlendShape sh = blend_shape(QW, QH);
    uint32_t tpw_log2 = p.tpw_log2 >= 0 ? (uint32_t)p.tpw_log2 : blend_tpw_log2(p.tiles_x, p.tiles_y, sh);
    if (tpw_log2 > sh.tbx_log2 + sh.tby_log2) tpw_log2 = sh.tbx_log2 + sh.tby_log2;
    const uint32_t grid = blend_grid_blocks(p.tiles_x, p.tiles_y, sh, tpw_log2);
    constexpr int NT = 64 * QW * QH;
    const bool capture = p.debug_consumed != nullptr || p.debug_walked != nullptr;  // analysis build of the kernel
    // tuning knob (WS_BLEND_LDS_PAD_KB): unused dynamic LDS that lowers the number of blend workgroups per CU
    const size_t pad = (size_t)p.lds_pad_kb * 1024u;
    if (p.debug_timing) {  // analysis: the production form (one tile per workgroup, f32 target, 32x32) with time stamps
        if (QW != 4 || QH != 4 || p.format != WS_FORMAT_RGBA32_FLOAT || capture || tpw_log2 > 0u || p.dma)
            return fail(WS_ERR_UNSUPPORTED, "blend timing: 32x32 tiles, rgba32float target, one tile per workgroup, no capture / DMA");
        hipLaunchKernelGGL((k_blend<WS_FORMAT_RGBA32_FLOAT, 4, 4, false, false, false, true>), dim3(grid), dim3(1024), pad, stream, p,
                           tpw_log2);
        WS_HIP(hipGetLastError());
        return WS_OK;
    }
#ifdef WS_EXPERIMENTAL
    // the barrier-free form (k_blend2): one 32x32 tile per workgroup, production launch only
    if (p.async_staging && QW == 4 && QH == 4 && !capture && tpw_log2 == 0u && !p.dma && p.range_row_shift == 0u) {
        switch (p.format) {
            case WS_FORMAT_RGBA32_FLOAT: hipLaunchKernelGGL(k_blend2<WS_FORMAT_RGBA32_FLOAT>, dim3(grid), dim3(1024), pad, stream, p); break;
            case WS_FORMAT_RGBA16_FLOAT: hipLaunchKernelGGL(k_blend2<WS_FORMAT_RGBA16_FLOAT>, dim3(grid), dim3(1024), pad, stream, p); break;
            case WS_FORMAT_RGBA8_UNORM: hipLaunchKernelGGL(k_blend2<WS_FORMAT_RGBA8_UNORM>, dim3(grid), dim3(1024), pad, stream, p); break;
            default: return fail(WS_ERR_INVALID, "blend: unknown colour format");
        }
        WS_HIP(hipGetLastError());
        return WS_OK;
    }
#else
    if (p.async_staging) return fail(WS_ERR_UNSUPPORTED, "barrier-free staging (k_blend2) is only in the experimental build");
#endif
#ifdef WS_EXPERIMENTAL  // (LDS-DMA staging, WS_BLEND_DMA=1: measured neutral; instantiated in the experimental build only)
#define WS_LAUNCH_BLEND_DMA(FMT)                                                                                          \
    if (!capture && tpw_log2 > 0u && p.dma)                                                                               \
        hipLaunchKernelGGL((k_blend<FMT, QW, QH, true, false, true>), dim3(grid), dim3(NT), pad, stream, p, tpw_log2);    \
    else if (!capture && p.dma)                                                                                           \
        hipLaunchKernelGGL((k_blend<FMT, QW, QH, false, false, true>), dim3(grid), dim3(NT), pad, stream, p, tpw_log2);   \
    else
#else
#define WS_LAUNCH_BLEND_DMA(FMT)
    if (p.dma) return fail(WS_ERR_UNSUPPORTED, "LDS-DMA staging is only in the experimental build");
#endif
#define WS_LAUNCH_BLEND(FMT)                                                                                              \
    WS_LAUNCH_BLEND_DMA(FMT)                                                                                              \
    if (p.exact_cut && !capture && tpw_log2 > 0u)                                                                         \
        hipLaunchKernelGGL((k_blend<FMT, QW, QH, true, false, false, false, true>), dim3(grid), dim3(NT), pad, stream, p, tpw_log2);  \
    else if (p.exact_cut && !capture)                                                                                     \
        hipLaunchKernelGGL((k_blend<FMT, QW, QH, false, false, false, false, true>), dim3(grid), dim3(NT), pad, stream, p, tpw_log2); \
    else if (capture)                                                                                                          \
        hipLaunchKernelGGL((k_blend<FMT, QW, QH, true, true, false>), dim3(grid), dim3(NT), pad, stream, p, tpw_log2);    \
    else if (tpw_log2 > 0u)                                                                                               \
        hipLaunchKernelGGL((k_blend<FMT, QW, QH, true, false, false>), dim3(grid), dim3(NT), pad, stream, p, tpw_log2);   \
    else                                                                                                                  \
        hipLaunchKernelGGL((k_blend<FMT, QW, QH, false, false, false>), dim3(grid), dim3(NT), pad, stream, p, tpw_log2)
    switch (p.format) {
        case WS_FORMAT_RGBA32_FLOAT:
            WS_LAUNCH_BLEND(WS_FORMAT_RGBA32_FLOAT);
            break;
        case WS_FORMAT_RGBA16_FLOAT:
            WS_LAUNCH_BLEND(WS_FORMAT_RGBA16_FLOAT);
            break;
        case WS_FORMAT_RGBA8_UNORM:
            WS_LAUNCH_BLEND(WS_FORMAT_RGBA8_UNORM);
            break;
        default:
            return fail(WS_ERR_INVALID, "blend: unknown colour format");
    }
#undef WS_LAUNCH_BLEND
#undef WS_LAUNCH_BLEND_DMA
    WS_HIP(hipGetLastError());
    return WS_OK;
}

int launch_blend(const BlendParams& p, int variant, hipStream_t stream) {
    const uint32_t ntiles = p.tiles_x * p.tiles_y;
    if (ntiles == 0) return WS_OK;
    if (variant == 2) {  // WS_BLEND_TARGET_PRECISION: back to front, destination rounded after every splat
        const uint32_t groups = ((ntiles + 7u) / 8u) * 8u * p.qw * p.qh;
        switch (p.format) {
            case WS_FORMAT_RGBA32_FLOAT:
                hipLaunchKernelGGL(k_blend_strict<WS_FORMAT_RGBA32_FLOAT>, dim3(groups), dim3(64), 0, stream, p);
                break;
            case WS_FORMAT_RGBA16_FLOAT:
                hipLaunchKernelGGL(k_blend_strict<WS_FORMAT_RGBA16_FLOAT>, dim3(groups), dim3(64), 0, stream, p);
                break;
            case WS_FORMAT_RGBA8_UNORM:
                hipLaunchKernelGGL(k_blend_strict<WS_FORMAT_RGBA8_UNORM>, dim3(groups), dim3(64), 0, stream, p);
                break;
            default:
                return fail(WS_ERR_INVALID, "blend: unknown colour format");
        }
        WS_HIP(hipGetLastError());
        return WS_OK;
    }
#ifdef WS_EXPERIMENTAL
    if (variant == 1) {  // one wave per 8x8 quadrant, no LDS (cross-check)
        const uint32_t groups = ((ntiles + 7u) / 8u) * 8u * p.qw * p.qh;
        switch (p.format) {
            case WS_FORMAT_RGBA32_FLOAT:
                hipLaunchKernelGGL(k_blend_q<WS_FORMAT_RGBA32_FLOAT>, dim3(groups), dim3(64), 0, stream, p);
                break;
            case WS_FORMAT_RGBA16_FLOAT:
                hipLaunchKernelGGL(k_blend_q<WS_FORMAT_RGBA16_FLOAT>, dim3(groups), dim3(64), 0, stream, p);
                break;
            case WS_FORMAT_RGBA8_UNORM:
                hipLaunchKernelGGL(k_blend_q<WS_FORMAT_RGBA8_UNORM>, dim3(groups), dim3(64), 0, stream, p);
                break;
            default:
                return fail(WS_ERR_INVALID, "blend: unknown colour format");
        }
        WS_HIP(hipGetLastError());
        return WS_OK;
    }
#else
    if (variant == 1) return fail(WS_ERR_UNSUPPORTED, "blend variant 1 (k_blend_q) is only in the experimental build");
#endif
    if (p.qw == 2u && p.qh == 2u) return launch_blend_shape<2, 2>(p, stream);
    if (p.qw == 4u && p.qh == 2u) return launch_blend_shape<4, 2>(p, stream);
    if (p.qw == 4u && p.qh == 4u) return launch_blend_shape<4, 4>(p, stream);
    return fail(WS_ERR_INVALID, "blend: unsupported tile shape");
}

// host-side twin of the staging step (CPU unit test of the quadrant mask; not on any render path)
int debug_stage_splat(const uint32_t w[5], float W, float H, float tile_x0, float tile_y0, uint32_t qw, uint32_t qh,
                      float rec[10], uint32_t* mask) {
    stage::Staged s;
    if (qw == 2u && qh == 2u) s = stage::decode<2, 2>(w[0], w[1], w[2], w[3], w[4], W, H, tile_x0, tile_y0, CUT_A * stage::LOG2E_F);
    else if (qw == 4u && qh == 2u) s = stage::decode<4, 2>(w[0], w[1], w[2], w[3], w[4], W, H, tile_x0, tile_y0, CUT_A * stage::LOG2E_F);
    else if (qw == 4u && qh == 4u) s = stage::decode<4, 4>(w[0], w[1], w[2], w[3], w[4], W, H, tile_x0, tile_y0, CUT_A * stage::LOG2E_F);
    else return fail(WS_ERR_INVALID, "unsupported tile shape");
    const float v[10] = {s.i00, s.i01, s.c0, s.i10, s.i11, s.c1, s.alpha, s.r, s.g, s.b};
    for (int i = 0; i < 10; ++i) rec[i] = v[i];
    *mask = s.mask;
    return WS_OK;
}

// host-side twin of the binning footprint (footprint.h; CPU unit test, not on any render path): the tile ids the kept
// ellipse of the splat reaches, in emission order
int debug_footprint(const uint32_t w[3], float vw, float vh, uint32_t tile_w_log2, uint32_t tile_h_log2, uint32_t tiles_x,
                    uint32_t capacity, uint32_t* tiles, uint32_t* count) {
    const fp::Tiles ft = fp::setup(w[0], w[1], w[2], vw, vh, tile_w_log2, tile_h_log2);
    const uint32_t n = fp::count(ft, tile_w_log2, tile_h_log2);
    *count = n;
    for (uint32_t k = 0; k < n && k < capacity; ++k) tiles[k] = fp::tile_at(ft, k, tiles_x, tile_w_log2, tile_h_log2);
    return WS_OK;
}

}  // namespace ws
