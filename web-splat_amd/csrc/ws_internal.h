// ws_internal.h -- shared internal declarations of libwebsplat_hip (not part of the ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <stdint.h>

#include <string>

#include "websplat.h"

#if defined(_OPENMP) && !defined(__HIP_DEVICE_COMPILE__)
#include <omp.h>
// kmp_get_blocktime / kmp_set_blocktime are extensions of LLVM's (and Intel's) OpenMP runtime: only there
#if defined(__clang__) && defined(KMP_VERSION_MAJOR)
#define WS_HAVE_KMP_BLOCKTIME 1
#endif
#endif

namespace ws {
// Stride of the per-frame Splat records (pointcloud.rs:352-358: 20 B of f16 values).  32 pads every record to its own 32-B
// sector: the blend's random gathers then fetch one sector per record instead of 1.5 (A/B: -DWS_SPLAT_STRIDE=32).
#ifndef WS_SPLAT_STRIDE
#define WS_SPLAT_STRIDE 20
#endif
constexpr uint32_t SPLAT_STRIDE = WS_SPLAT_STRIDE;
static_assert(SPLAT_STRIDE == 20 || SPLAT_STRIDE == 32, "Splat records are 20 B, optionally padded to 32");


// The library's host loops (scene re-layout, PLY row conversion) are OpenMP regions.  LLVM's OpenMP runtime keeps the
// workers of a finished region SPINNING for KMP_BLOCKTIME -- 200 ms by default -- before they sleep; on a 128-thread host
// that is 128 busy cores next to the thread that enqueues frames and to the HIP runtime's own threads.  Measured on the
// MI355X box (scripts/slowmode_probe*.py): for ~0.2 s after a point cloud was created on another renderer's watch, four
// frames in flight ran at 2 600 instead of 16 800 frames/s -- the "slow cells" of the N x resolution sweeps of rounds 2
// and 3.  Every host function with a parallel region holds one of these: its workers go to sleep as the region ends.
// Side effect: the block time is a per-thread control variable of the CALLING thread; a parallel region the same thread
// starts concurrently (it cannot: the guard lives on its stack) would see 0.  Other threads' regions are unaffected.  With
// another OpenMP runtime (libgomp) the guard is a no-op.
struct OmpQuietWorkers {
    int saved = 0;
#if defined(WS_HAVE_KMP_BLOCKTIME)
    OmpQuietWorkers() : saved(kmp_get_blocktime()) { kmp_set_blocktime(0); }
    ~OmpQuietWorkers() { kmp_set_blocktime(saved); }
#else
    OmpQuietWorkers() {}
    ~OmpQuietWorkers() {}
#endif
};

// ---- error plumbing -------------------------------------------------------------------------
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
int hip_fail(hipError_t e, const char* what);

#define WS_HIP(expr)                                          \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) return ::ws::hip_fail(_e, #expr); \
    } while (0)

// ---- geometry of the rasteriser ----------------------------------------------------------------
// Binning tiles are QW x QH quadrants of 8x8 pixels (one wave each): 2x2 = the north star's 16x16 tile; 4x2 / 4x4
// bin two / four such tiles into one list (ws_context::tile_qw/qh, WS_TILE_SHAPE).  Default 4x4, by measurement:
// a 24-px splat touches 8.3 16x16 tiles but 3.7 32x32 ones, so the (tile, splat) entry list -- emit, tile sort,
// gather -- shrinks 2.2x while the per-quadrant compositing work is unchanged.
constexpr int QUAD = 8;
constexpr float CUTOFF = 2.3539888583335364f; // gaussian.wgsl:2  sqrt(ln 255)
constexpr float CUT_A = 2.0f * CUTOFF;        // gaussian.wgsl:61 discard if a > 2*CUTOFF
constexpr float T_MIN = 1.0f / 16384.0f;      // front-to-back early-out (6.1e-5; DESIGN.md section Blend)

// ---- device-side frame state, zeroed by ONE memset at the start of every frame ------------------
// Wave priority (s_setprio 0..3) of K1 and of the frame's small dependent kernels against the blend's default 0: which wave a
// SIMD issues from when several are ready.  A/B switches (round 6, verdict r05 item 1); 0 = no instruction emitted.
#ifndef WS_PRIO_K1
#define WS_PRIO_K1 0
#endif
#ifndef WS_PRIO_SMALL
#define WS_PRIO_SMALL 0
#endif
#define WS_SETPRIO_K1() do { if (WS_PRIO_K1) __builtin_amdgcn_s_setprio(WS_PRIO_K1); } while (0)
#define WS_SETPRIO_SMALL() do { if (WS_PRIO_SMALL) __builtin_amdgcn_s_setprio(WS_PRIO_SMALL); } while (0)
// ---- device-side launch trace (ws_renderer_enable_frame_trace): min of the workgroups' start stamps, max of their end stamps ----
#ifdef __HIPCC__
__device__ __forceinline__ unsigned long long ws_realtime() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ void ws_trace_begin(unsigned long long* t) {
    if (t && threadIdx.x == 0) atomicMin(t, ws_realtime());
}
__device__ __forceinline__ void ws_trace_end(unsigned long long* t) {
    if (t && threadIdx.x == 0) atomicMax(t + 1, ws_realtime());
}
#endif
#ifndef WS_BLEND_ASYNC_DEFAULT
#define WS_BLEND_ASYNC_DEFAULT 0
#endif
#ifndef WS_DEPTH_DIGIT_BITS_DEFAULT
#define WS_DEPTH_DIGIT_BITS_DEFAULT 8
#endif
#ifndef WS_DEPTH_DIGIT_BITS_ADAPTIVE   // 1: a renderer picks 8 or 9 bits per frame from its previous frame's key range (ws_api.cpp)
#define WS_DEPTH_DIGIT_BITS_ADAPTIVE 1
#endif
constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;

struct FrameCounters {
    uint32_t num_visible;    // V  (reference: SortInfos.keys_size, preprocess.wgsl:262)
    uint32_t k1_ticket;      // dynamic block id dispenser of the preprocess kernel
    uint32_t num_entries;    // D  (clamped to capacity)
    uint32_t overflow;       // bit 0: D exceeded capacity; bits 1..3: a look-back spin timed out (K1/bin/sort)
    uint32_t sort_ticket[8]; // [0..3] chunk dispensers of the fat-tile depth sort's passes (k_dsort_fat); [4..7] spare
    uint32_t bin_ticket;     // block dispenser of the binning prefix kernel
    uint32_t entries_needed; // D before clamping to the capacity (what a retry has to allocate)
    uint32_t epoch;          // look-back epoch of the frame, written by K1 (the only kernel whose arguments change per
                             // frame: later kernels read it here, so that a captured frame graph replays with ONE update)
    uint32_t bin_request;    // written by K1: BIN_NEVER / BIN_AUTO / BIN_ALWAYS (the frame's request for coarse binning)
    // --- second cache line: what the frame decided ---
    uint32_t bin_shift;      // written by k_bin_prefix: 0 = the frame binned at the blend's tile size, 1 = at twice that
    uint32_t depth_skip_top; // written by the depth sort's first kernel: 1 = the frame's keys span less than 2^24 above
                             // depth_key_base: the sort's fourth digit pass has nothing to do, the result is where pass 2 left it
    uint32_t depth_key_base; // written with it: passes 1..3 take their digits from (key - depth_key_base) (depth_range_decide)
    uint32_t depth_span_class; // written with them: 0 = no key reported, 1 = the frame's keys span < 2^24 above their 256-aligned minimum (three
                               // 8-bit passes sort them), 2 = they do not (three 9-bit passes do, below 2^27).  The blend posts it to the host:
                               // the NEXT frame of the renderer picks its digit width by it (ws_api.cpp; either width gives the same order)
    uint32_t _pad[12];
    // --- the frame's footprint totals, summed by K1: TILE_SUM_SLOTS x {sum at the blend's tile size, sum at twice that
    // size}, one 64-B line per slot.  Workgroup b adds its partial sums to slot b % 16 with two returnless atomics; all
    // K1 workgroups retire within a few microseconds of each other, and ~1000 atomics on ONE address drain one after
    // the other (~12 ns each: measured +13 us per frame), on 16 lines in 0.7 us.  k_bin_prefix adds the slots (bin_shift_decide).
    // words [2], [3] of a slot: the range of the frame's depth keys -- max(~key) (= ~min; 0 = nothing reported) and max(key),
    // left by the depth sort's first histogram kernel, two returnless atomics per sort tile (depth_range_decide)
    uint32_t tile_sums[16 * 16];
};
constexpr int TILE_SUM_SLOTS = 16;
constexpr int TILE_SUM_STRIDE = 16;  // uint32 words per slot (64 B)
static_assert(sizeof(FrameCounters) == 128 + TILE_SUM_SLOTS * TILE_SUM_STRIDE * 4, "FrameCounters layout");

// Everything a frame needs zeroed lives in one contiguous arena: counters, the digit histograms of both
// sorts, and the tile ranges (appended after this struct).
struct FrameZero {
    FrameCounters counters;
    uint32_t depth_hist[4 * 512];   // digit totals of the depth sort's passes (8-bit digits use the first 256 of every row of 512)
    uint32_t tile_hist[4 * RADIX];
    uint32_t fat_barrier[9 * 16];  // single-launch depth sort (WS_DEPTH_SORT=coop): state of its device-wide barriers (grid_barrier.h)
    // uint2 tile_ranges[tiles] follows
};

// ---- binning footprint of a splat: the companion value of the depth sort ----------------------------------------
// K1 stores one 32-bit word per visible splat that says which binning tiles the splat is listed in; it rides through the
// depth sort with the splat index (sort.hip), so the binning prefix streams it in draw order.  Three forms:
//   FP_RECT_PACKED (default)  the bounding rectangle of the kept ellipse, x0 | y0 << 8 | (w - 1) << 16 | (h - 1) << 24
//                  in binning tiles: at most 256 tiles per axis (8192 px at the default 32-px tile); RECT_EMPTY = no
//                  tile.  k_bin_emit derives a tile from the word alone.
//   FP_RECT_COUNT  the NUMBER of tiles of the same rectangle; k_bin_emit re-derives the rectangle from the 12 geometry
//                  bytes of the Splat record (footprint.h).  Taken automatically for viewports beyond 256 tiles per axis:
//                  any target the reference can create (it asks for the adapter's own max_texture_dimension_2d,
//                  src/lib.rs:99-110) is accepted, at the price of a slower emit kernel.
//   FP_ELLIPSE     the number of tiles the kept ELLIPSE reaches (per tile row the exact column span, footprint.h):
//                  15 % fewer entries on the 1 M / 1080p scene, but the span arithmetic costs more VALU time in K1 and
//                  k_bin_emit than the entries cost downstream (profiles/r03/footprint_*) -- WS_FOOTPRINT=ellipse.
enum FootprintMode { FP_RECT_PACKED = 0, FP_RECT_COUNT = 1, FP_ELLIPSE = 2 };
constexpr uint32_t RECT_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t RECT_PACKED_MAX_TILES_PER_AXIS = 256u;
constexpr uint32_t MAX_TILES_PER_AXIS = 65535u;  // tile coordinates are 16-bit (the blend packs tx | ty << 16)
__host__ __device__ inline uint32_t rect_pack(uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) {
    return x0 | (y0 << 8) | ((x1 - x0) << 16) | ((y1 - y0) << 24);
}
__host__ __device__ inline uint32_t rect_tiles(uint32_t r) {
    return r == RECT_EMPTY ? 0u : (((r >> 16) & 0xFFu) + 1u) * ((r >> 24) + 1u);
}

// ---- binning granularity, decided PER FRAME ON THE DEVICE -----------------------------------------------------------
// The blend composites 32x32-px tiles (default shape).  Binning may use those tiles, or 2 x 2 blocks of them (64x64 px):
// four blend workgroups then share one binned list -- half the (tile, splat) entries to emit and sort when splats span
// several tiles, at the price of every entry being staged by up to four workgroups.  Measured (profiles/r03/bin64_*):
// +16..+20 % frames/s where the rectangles shrink 2.07x (hd1m, c4), +3..+5 % at 1.75x (c2), -14 % / -23 % at 1.14x / 1.10x
// (c5, c3: pixel-sized splats); an N x resolution sweep (profiles/r03/sweep_bin_auto_v24.json) has 64 px ahead (+1..+63 %) at
// all ratios 1.75..2.63 and by +4 % on average (-3..+13 %) at 1.53: the threshold sits just below the smallest ratio measured to win.  K1 sums both tile counts while it writes the rectangles; the binning prefix derives
// the decision from the two sums (a pure function of the frame: ranks and renderers agree), no host round trip.
enum BinRequest { BIN_NEVER = 0, BIN_AUTO = 1, BIN_ALWAYS = 2 };
#ifndef WS_BIN64_RATIO_PERCENT
#define WS_BIN64_RATIO_PERCENT 150
#endif
constexpr uint32_t BIN64_RATIO_PERCENT = WS_BIN64_RATIO_PERCENT;  // coarse binning when the sum at the blend's tile size >= 1.50 x the sum at twice that
__host__ __device__ inline uint32_t rect_tiles64(uint32_t r) {
    if (r == RECT_EMPTY) return 0u;
    const uint32_t x0 = r & 0xFFu, y0 = (r >> 8) & 0xFFu, x1 = x0 + ((r >> 16) & 0xFFu), y1 = y0 + (r >> 24);
    return ((x1 >> 1) - (x0 >> 1) + 1u) * ((y1 >> 1) - (y0 >> 1) + 1u);
}
// the packed rectangle in units of 2 x 2 tiles
__host__ __device__ inline uint32_t rect_coarse(uint32_t r) {
    const uint32_t x0 = r & 0xFFu, y0 = (r >> 8) & 0xFFu, x1 = x0 + ((r >> 16) & 0xFFu), y1 = y0 + (r >> 24);
    return rect_pack(x0 >> 1, y0 >> 1, x1 >> 1, y1 >> 1);
}
// 0 = bin at the blend's tile size, 1 = at twice that.  k_bin_prefix -- the first kernel that needs the answer -- derives it
// in every workgroup and leaves it in counters->bin_shift for the kernels behind it and for the host (32 loads per
// workgroup were measurable in the blend: -2 % frames/s on c2 / c5).
// The depth keys are bits(zfar - z) of the visible splats (preprocess.wgsl:270-273), sorted as u32 by four 8-bit LSD passes
// (gpu_rs.rs:865-884).  On a scene the camera sees from outside the keys of a frame span far less than 32 bits -- c3: depths
// 4.0 .. 14.9, i.e. 0x40800000 .. 0x416EE000, a range of 2^23.9 -- and sorting (key - base) for any base <= min(key) gives the
// same stable order.  The sort's FIRST kernel (the per-tile histogram of pass 0, which reads every key anyway) leaves max(~key) and
// max(key) in the slots below; the column scan behind it folds them: base = min(key) with its low byte cleared (so that the
// first pass's digit, key & 0xFF, is the digit of key - base: that pass runs before anybody knows the base), and when max(key) - base < 2^24 the fourth pass is a pass over
// a constant digit -- the identity: its three kernels leave at once, and the readers of the sorted arrays (k_bin_prefix,
// k_bin_emit, the host's read-back) take them from where pass 2 left them.  Three launches and 24 of 112 B per key less on
// such frames; order and stability are the reference's.  -> (base, skip) in the counters
// `digits` = 2^b, the radix of the sort: b = 8 is the reference's shape (four passes, the fourth skipped below 2^24); b = 9
// (round 6) is three passes over key - base whenever the frame's keys span less than 2^27 -- every frame of every BASELINE
// workload -- with a fourth pass over bits 27..31 enqueued for the frames that do not, which leaves at once on all others.
// A frame that holds a key of 0xFFFFFFFF (the compressed path's saturating u32(f32), bits(NaN)) keeps base = 0: the scatter
// kernels exempt that value from the subtraction (it is also their padding key), the histogram kernels do not.
__host__ __device__ inline void depth_range_decide(FrameCounters* c, uint32_t digits = 256u) {
    uint32_t not_min = 0u, mx = 0u;
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) {
        const uint32_t a = c->tile_sums[sl * 16 + 2], b = c->tile_sums[sl * 16 + 3];
        not_min = a > not_min ? a : not_min;
        mx = b > mx ? b : mx;
    }
    const bool known = not_min != 0u;          // (no key reported: nothing visible, or a K1 form that does not report)
    const uint32_t base = (known && mx != 0xFFFFFFFFu) ? (~not_min & ~(digits - 1u)) : 0u;
    const uint32_t span = digits * digits * digits;  // 2^24 / 2^27: what three passes cover
    c->depth_key_base = base;
    c->depth_skip_top = (known && mx >= base && (mx - base) < span) ? 1u : 0u;
    const uint32_t base8 = (known && mx != 0xFFFFFFFFu) ? (~not_min & ~255u) : 0u;
    c->depth_span_class = !known ? 0u : ((mx - base8) < (1u << 24) ? 1u : 2u);
}

__host__ __device__ inline uint32_t bin_shift_decide(const FrameCounters* c) {
    const uint32_t req = c->bin_request;
    if (req == BIN_ALWAYS) return 1u;
    if (req != BIN_AUTO) return 0u;
    uint32_t sum32 = 0u, sum64 = 0u;  // (sums of at most 2^32 / 2^16 splats' tile counts would wrap: the ratio of a frame
#pragma unroll                        //  that large is meaningless either way, and every reader wraps alike)
    for (int sl = 0; sl < TILE_SUM_SLOTS; ++sl) {
        sum32 += c->tile_sums[sl * TILE_SUM_STRIDE];
        sum64 += c->tile_sums[sl * TILE_SUM_STRIDE + 1];
    }
    return ((uint64_t)sum32 * 100ull >= (uint64_t)sum64 * BIN64_RATIO_PERCENT && sum64 != 0u) ? 1u : 0u;
}


// ---- kernel parameter blocks (passed by value; the analogue of the reference's uniform buffers) ---
struct K1Params {
    ws_camera_uniform cam;   // renderer.rs:290-306
    ws_settings_uniform rs;  // renderer.rs:602-618
    ws_gaussian_quantization quant; // compressed only (pointcloud.rs:389-396)
    uint32_t num_points;
    uint32_t sh_deg_layout;  // compressed: number of coefficients per packed SH record, (sh_deg+1)^2
    uint32_t tiles_x, tiles_y;
    uint32_t tile_w_log2, tile_h_log2;  // tile size in pixels (8 * QW, 8 * QH)
    uint32_t epoch;          // look-back epoch of this frame (lookback.h)
    float znear, zfar;       // -proj[3][2]/proj[2][2], -proj[3][2]/(proj[2][2]-1)  (preprocess.wgsl:270-271)
    uint32_t fade_done;      // walltime is past every Gaussian's fade-in: scale_mod == 1 exactly
    uint32_t bin_request;    // BinRequest of this frame (coarse binning never / decided on the device / always)
};

// uncompressed point cloud in HBM: eight planes of 16-B chunks, plane p of Gaussian i at
// planes + (p*N + i)*16.  plane 0 = {x,y,z, opacity f16 | pad}; plane 1 = {cov f16 x6, pad 4 B};
// planes 2..7 = the 96-B SH record ([[f16;3];16]) in 16-B pieces.
constexpr int PC_PLANES = 8;

// ---- per-launch GPU timestamps (the reference's GPUStopwatch, utils.rs:26-134, at kernel granularity) --------
// HIP events recorded on the launch stream between the kernels of one frame; only when the caller asked for
// per-kernel times (ws_renderer_enable_timers(r, 2)): the events themselves perturb a throughput run.
struct KernelMarks {
    static constexpr int MAX = 48;
    hipEvent_t ev[MAX + 1] = {};
    const char* label[MAX] = {};
    int n = 0;
    bool created = false;
    bool active = false;
    hipStream_t stream = nullptr;
    int create();
    void destroy();
    void begin(hipStream_t s, bool restart);
    void mark(const char* what);
};
inline void km_mark(KernelMarks* km, const char* what) {
    if (km && km->active) km->mark(what);
}

// ---- sort ---------------------------------------------------------------------------------------
constexpr int SORT_THREADS = 256;
#ifndef WS_SORT_KPT
#define WS_SORT_KPT 8
#endif
constexpr int SORT_KPT = WS_SORT_KPT;                 // keys per thread
constexpr int SORT_TILE = SORT_THREADS * SORT_KPT;    // keys per work tile
constexpr int SORT_KPT_SMALL = 4;                     // small inputs: 1024-key tiles -> 4x the workgroups
#ifndef WS_SORT_SMALL_MAX
#define WS_SORT_SMALL_MAX (2u << 20)
#endif
constexpr uint32_t SORT_SMALL_MAX = WS_SORT_SMALL_MAX;  // host-side bound n up to which the small tile is used
uint32_t sort_tile_size(uint32_t n);                  // tile size the scan path (the production sort) uses for bound n

// Single-pass tile-id sort (launch_tile_sort_wide): the whole tile id is ONE digit of up to 11 bits.
constexpr int TILE_SORT_WIDE_MAX_BITS = 11;
constexpr int TILE_SORT_WIDE_MAX_BINS = 1 << TILE_SORT_WIDE_MAX_BITS;

struct SortScratch {
    uint32_t* keys_alt = nullptr;     // ping-pong partner of the caller's key buffer   [cap]
    uint32_t* vals_alt = nullptr;     // ping-pong partner of the caller's value buffer [cap]
    uint32_t* hist = nullptr;         // [4][hist_pitch] digit totals of the passes (written by the column scans)
    uint32_t hist_pitch = 256;        // words per pass in hist (512 for a scratch that may run 9-bit digits)
    uint32_t* tile_sums = nullptr;    // [rows][tiles_cap] per-tile digit counts / offsets, rows = 256 (512: 9-bit digits)
    uint32_t rows = 256;
    uint32_t cap = 0;
    uint32_t tiles = 0;               // ceil(cap / SORT_TILE)
    uint32_t tiles_cap = 0;           // row pitch of tile_sums: the largest tile count any n <= cap can need
    uint32_t wide_bins = 0;           // != 0: tile_sums also holds [tiles][wide_bins] rows (launch_tile_sort_wide)
    uint32_t* wide_hist = nullptr;    // [wide_bins] totals per bin, written by the wide column scan
};

// Launch an ascending stable LSD radix sort of (key, value) pairs on `stream`.
//   d_count == nullptr -> sort n pairs; else the count is read on the device (clamped to n).
//   begin_bit/end_bit: key bits that participate; digit_bits (6..8): digit width, passes = ceil(bits / digit_bits).
//   implicit_iota: values of the first pass are the element positions (vals in is not read).
//   first_tile_hist_ready: the producer of the keys already wrote the first pass's per-tile digit
//     counts into sc.tile_sums (layout [digit][tiles_cap], tile = sort_tile_size(n) consecutive keys).
//   ranges != nullptr: the LAST pass does not write the sorted keys; instead ranges[key] (key < nranges, zero on
//     entry) receives (0xFFFFFFFF - begin, end) of that key's run in the sorted order (see k_sort_scatter).
// The result lands in (keys, vals) if the pass count is even, else in (scratch.keys_alt, vals_alt);
// *out_keys / *out_vals receive the final pointers.
int launch_sort_pairs(const SortScratch& sc, uint32_t* keys, uint32_t* vals, const uint32_t* d_count, uint32_t n,
                      int begin_bit, int end_bit, bool implicit_iota, bool first_tile_hist_ready, hipStream_t stream,
                      uint32_t** out_keys, uint32_t** out_vals, KernelMarks* km = nullptr, const char* tag = "", uint2* ranges = nullptr, uint32_t nranges = 0,
                      int digit_bits = RADIX_BITS, bool key16 = false, uint32_t* aux = nullptr, uint32_t* aux_alt = nullptr,
                      FrameCounters* skip_top = nullptr, uint32_t** out_keys_skipped = nullptr, uint32_t** out_vals_skipped = nullptr,
                      int kpt9 = 0);  // kpt9 (9-bit digits only): 0 = the tile size the input size selects, 4 / 8 = keys per thread (A/B)
//   skip_top (four passes of 8- or 9-bit digits from bit 0 only: the depth sort of a frame): the first histogram kernel leaves the key range, the column scan behind it folds it into
//     skip_top->depth_skip_top (depth_range_decide); when it is set the three kernels of the LAST pass return at once and the
//     result is what pass 2 wrote: *out_keys_skipped / *out_vals_skipped (the companion values next to the payload).
//   aux / aux_alt: a 4-byte companion value per pair travels with the payload; the result lands where
//     the payload lands (aux for an even pass count, aux_alt for an odd one).

// ---- single-pass tile-id sort (sort.hip) ---------------------------------------------------------------------------
// Stable counting sort of n (16-bit tile id, value) pairs whose ids are below `bins` = 2^bits <= 2048, in TWO launches:
// the producer (k_bin_emit<., WIDE>) has left the per-tile bin counts in sc.tile_sums as [tile][bins] (tile = SORT_TILE
// consecutive pairs); k_tile_col_scan_wide turns every column into exclusive offsets and writes the bin totals, and
// k_tile_scatter_wide ranks each tile's pairs per bin (ballot match, as the digit passes do) and writes the VALUES only,
// to sc.vals_alt.  The ranges of the ids in the sorted order are prefix sums of the totals: ranges[id] = (0xFFFFFFFF -
// begin, end) for non-empty ids (zero on entry), no atomics.  Against the two 6-bit passes it replaces: 2 launches instead
// of 5, the pairs are read once instead of twice and never written back as pairs -- but the count rows are 4 B x bins per
// 2048 pairs, at 2048 bins as many bytes as the pairs: faster alone, slower with frames in flight (WS_TILE_SORT=wide).
int launch_tile_sort_wide(const SortScratch& sc, const uint32_t* keys16, const uint32_t* vals, const uint32_t* d_count,
                          uint32_t n, int bits, hipStream_t stream, KernelMarks* km, uint2* ranges, uint32_t nranges);

// ---- fat-tile one-sweep depth sort (sort.hip k_dsort_fat) -----------------------------------------------------------
// The same result as launch_sort_pairs over bits 0..32 with a companion value: four 8-bit passes, A -> B -> A -> B -> A, the
// sorted arrays end where they started.  At most FAT_MAX_GRID workgroups of 1024 threads rank chunks of up to 8192 pairs;
// the cross-chunk prefix is a consumer-side sum over epoch-tagged count rows (no chain).  Per-pass launches (one histogram
// launch + four) or, coop = true, ONE launch with device-wide barriers between the passes (needs grid <= CUs resident).
constexpr uint32_t FAT_MAX_GRID = 256;
constexpr uint32_t FAT_CHUNK_MAX = 1024u * 8u;
struct FatSortScratch {
    uint32_t* keys_alt = nullptr;   // [cap] ping-pong partners
    uint32_t* vals_alt = nullptr;
    uint32_t* aux_alt = nullptr;
    uint32_t* hist = nullptr;       // [4][256] digit totals, zero on entry
    uint64_t* status = nullptr;     // [4][FAT_MAX_GRID][256] epoch-tagged chunk counts (zeroed once)
    uint32_t* tickets = nullptr;    // [4] chunk dispensers, zero on entry
    uint32_t* barrier = nullptr;    // 9 x 16 words, zero on entry (coop)
    const uint32_t* d_epoch = nullptr;  // != nullptr: the epoch is read from device memory (a captured frame graph replays)
    uint32_t* error = nullptr;      // OR-ed with 8 when a spin times out
    uint32_t cap = 0;
    int grid_request = 0;           // tuning (WS_DSORT_FAT_GRID): workgroups, 0 = automatic
};
size_t fat_sort_status_words();
// workgroups the fat sort uses for bound n on a part with num_cus CUs; 0 = n is beyond what this form handles (the caller
// takes the scan path)
uint32_t fat_sort_grid(uint32_t n, int num_cus, int grid_request);
int launch_depth_sort_fat(const FatSortScratch& sc, uint32_t* keys, uint32_t* vals, uint32_t* aux, const uint32_t* d_count,
                          uint32_t n, bool implicit_iota, bool coop, uint32_t epoch, int num_cus, hipStream_t stream,
                          KernelMarks* km = nullptr);

// ---- preprocess ---------------------------------------------------------------------------------
struct K1Buffers {
    const uint4* planes;         // uncompressed: PC_PLANES planes
    const uint8_t* gaussians_c;  // compressed: 24-B records
    const uint8_t* sh_bytes;     // compressed: packed int8 SH
    const uint8_t* covars;       // compressed: 12-B covariance codebook
    uint8_t* splats;             // [N] x 20 B  (pointcloud.rs:352-358 Splat)
    uint32_t* keys;              // [N] depth keys
    uint32_t* footprints;        // [N] the splat's binning footprint word (FootprintMode)
    uint32_t* src_index;         // [N] or nullptr (capture mode)
    uint64_t* block_status;      // [blocks] epoch-tagged look-back words
    FrameCounters* counters;
    unsigned long long* trace;   // nullptr, or {first workgroup start, last workgroup end} of this launch on the 100-MHz device clock
                                 //   (ws_renderer_enable_frame_trace: analysis of frames in flight; rocprofv3 serialises them)
};
int launch_preprocess(const K1Params& p, const K1Buffers& b, bool compressed, int footprint_mode, hipStream_t stream);
// K1 of up to K1_MAX_VIEWS views of ONE scene in one launch (preprocess.hip k_preprocess_multi): the scene is read once
constexpr int K1_MAX_VIEWS = 4;
struct K1MultiArgs {
    K1Params p[K1_MAX_VIEWS];
    K1Buffers b[K1_MAX_VIEWS];
    uint32_t nv;
};
int launch_preprocess_multi(const K1Params* p, const K1Buffers* b, uint32_t nv, bool compressed, int footprint_mode,
                            hipStream_t stream);
const void* preprocess_kernel_func(bool compressed, int footprint_mode);  // host-side kernel symbol (identifies K1's node in a captured graph)
uint32_t preprocess_blocks(uint32_t n);

// ---- binning + blend ----------------------------------------------------------------------------
constexpr int EMIT_TILE = SORT_TILE;  // tile entries produced per workgroup of the emit kernel (= the sort's tile)

struct BinBuffers {
    const uint32_t* sorted_idx;  // [V] store indices in draw order (far -> near)
    const uint32_t* fp_sorted;   // [N] footprint words by draw position (carried through the depth sort)
    const uint32_t* sorted_idx_alt;  // where both are when the depth sort's last pass was skipped (FrameCounters::depth_skip_top);
    const uint32_t* fp_sorted_alt;   //   nullptr: the sort cannot skip (compressed scenes sort 24 bits in three passes anyway)
    int footprint_mode;          // FootprintMode of those words
    const uint8_t* splats;       // [V] x 20 B Splat records (modes other than FP_RECT_PACKED: k_bin_emit re-derives the footprint from words 0..2)
    float vw, vh;                // viewport in pixels, as the camera uniform holds it
    uint32_t tile_w_log2, tile_h_log2;
    uint32_t* offsets;           // [N] exclusive prefix of tiles touched, by draw position
    uint32_t* emit_start;        // [cap / EMIT_TILE + 2] draw position owning entry m * EMIT_TILE
    uint64_t* block_status;      // look-back words of the prefix kernel
    uint32_t* entry_keys;        // [cap] tile ids
    uint32_t* entry_vals;        // [cap] store indices
    uint32_t* tile_hist;         // nullptr, or the tile sort's tile_sums: emit workgroup m also writes the digit
    uint32_t tile_hist_pitch;    //   counts (first digit of the tile id) of sort tile m -> no histogram pass 0
    uint32_t tile_hist_mask;     //   (1 << digit bits of the tile sort) - 1
    int tile_hist_wide;          // single-pass tile sort: tile_hist is [sort tile][bins], tile_hist_pitch = bins (<= 2048)
    int key16;                   // entry_keys holds uint16_t tile ids (fewer than 65535 tiles): 2 B less per entry
    uint32_t entry_cap;
    uint2* tile_ranges;          // [tiles] (begin, end) into the sorted entry list
    FrameCounters* counters;
    uint32_t max_points;         // N (upper bound of V)
    uint32_t tiles_x, tiles_y;
};
int launch_bin_prefix(const BinBuffers& b, hipStream_t stream);
int launch_bin_emit(const BinBuffers& b, hipStream_t stream);
uint32_t bin_prefix_blocks(uint32_t max_points);

struct BlendParams {
    const uint8_t* splats;      // [V] x 20 B
    const uint32_t* entry_vals; // sorted by tile, far -> near inside a tile
    const uint2* tile_ranges;   // (0xFFFFFFFF - begin, end) per tile, (0, 0) = empty  (written by the tile-id sort)
    uint32_t width, height, tiles_x, tiles_y;
    uint32_t qw, qh;            // quadrants (8x8 px, one wave each) per tile
    float background[4];
    void* out;
    size_t pitch;
    int format;
    int tpw_log2;               // log2(tiles per workgroup), -1 = automatic (blend_tpw_log2)
    int lds_pad_kb;             // tuning: extra (unused) dynamic LDS per workgroup, limits workgroups per CU
    int dma;                    // stage the Splat records with gfx950's LDS-DMA (global_load_lds) instead of through VGPRs
    int exact_cut;              // WS_BLEND_FAST_EXACT_CUT: the keep / discard decision of fragments at the cut-off on the oracle's expression
    int async_staging;          // k_blend2: double-buffered staging, LDS arrival counters instead of the two barriers per batch
    int num_cus;
    uint32_t bin_tiles_x;       // binning tiles per row at the blend's tile size (the frame may bin at twice that: FrameCounters::bin_shift)
    uint32_t range_row_shift;   // 0, or 1 = "split" mode: tiles_y counts HALF binning tiles (32x16 px, 8 waves) and the list of
                                //   blend tile (tx, ty) is the binning tile's (tx, ty >> 1): two workgroups share one list
    const FrameCounters* counters;  // this frame's counters: the error bits are folded into *sticky by the blend
    uint32_t* sticky;           // per-renderer error word that is NOT zeroed per frame (ws_renderer_errors)
    uint32_t* demand_mailbox;   // nullptr, or a host-visible (pinned, mapped) word: the entry demand of overflowed frames
    uint32_t* progress_mailbox; // nullptr, or a host-visible word: the blend's first workgroup posts frame_seq here when it starts
                                // (progress_mailbox[1]: the frame's depth_span_class, posted with it)
    uint32_t frame_seq;         //   (the frame's kernels in front of the blend are done: what a view batch bounds the host's run-ahead by)
    const uint4* order;         // nullptr, or [blockIdx] -> (tx | ty << 16, begin, end, -) longest list first (k_blend_order;
                                //   4x4 tiles, one tile per workgroup, not split)
    uint32_t* debug_consumed;   // nullptr, or [tiles]: entries of each tile's list the blend walked (capture mode)
    uint32_t* debug_walked;     // nullptr, or [tiles][17]: records walked per wave, [16] = sum over batches of the per-batch maximum
    uint32_t* debug_timing;     // nullptr, or [tiles][16 waves][BLEND_TIMING_WORDS]: per-wave phase times (ws_renderer_enable_blend_timing)
    unsigned long long* trace;  // nullptr, or {first workgroup start, last workgroup end} of this launch (ws_renderer_enable_frame_trace)
};
constexpr int BLEND_TIMING_WORDS = 16;
int launch_blend(const BlendParams& p, int variant, hipStream_t stream);
// the blend's workgroups in longest-list-first order (raster.hip k_blend_order): order[blend_order_blocks(..)]
uint32_t blend_order_blocks(uint32_t tiles_x, uint32_t tiles_y);
int launch_blend_order(const uint2* tile_ranges, const FrameCounters* counters, uint32_t tiles_x, uint32_t tiles_y, uint4* order,
                       int mode, hipStream_t stream);
int launch_empty(hipStream_t stream);
int debug_stage_splat(const uint32_t w[5], float W, float H, float tile_x0, float tile_y0, uint32_t qw, uint32_t qh,
                      float rec[10], uint32_t* mask);

int debug_footprint(const uint32_t w[3], float vw, float vh, uint32_t tile_w_log2, uint32_t tile_h_log2, uint32_t tiles_x,
                    uint32_t capacity, uint32_t* tiles, uint32_t* count);

// ---- PLY row decode on the GPU (ply_decode.hip) -------------------------------------------------------
int launch_ply_decode(const float* d_rows, uint32_t n, uint32_t sh_deg, uint4* planes, hipStream_t stream);

// ---- host math (host_math.cpp) ------------------------------------------------------------------
void build_camera_uniform(const ws_camera& cam, const uint32_t viewport[2], ws_camera_uniform* out);

// utils.rs:194-203 build_cov(rotation quaternion (s,x,y,z), scale) -> upper triangle xx,xy,xz,yy,yz,zz
void build_cov(const float q[4], const float scale[3], float out[6]);

// ---- half helpers ---------------------------------------------------------------------------------
uint16_t host_f32_to_f16(float f);
float host_f16_to_f32(uint16_t h);

}  // namespace ws

// grouped prepare of a view batch (ws_api.cpp): K1 once for up to K1_MAX_VIEWS renderers / views of one scene
struct ws_renderer;
struct ws_pointcloud;
int ws_internal_prepare_group(ws_renderer* const* rs, uint32_t n, const ws_pointcloud* pc, const ws_splatting_args* views,
                              hipStream_t const* streams);
// the largest tile-entry demand an overflowed frame of this renderer has left so far (0 = none), as last seen by
// ws_renderer_errors or a prepare() (ws_api.cpp)
uint32_t ws_internal_renderer_demand(const ws_renderer* r);
// Frames this renderer has enqueued (render() calls) and the number of the last one whose compositing kernel has STARTED on
// the device (posted by that kernel to pinned host memory: reading it costs a load, no runtime call).  A view batch keeps a
// slot's host side at most queue_depth frames ahead of it.  renderer_progress returns false when the renderer has no mailbox.
uint32_t ws_internal_renderer_frames_enqueued(const ws_renderer* r);
bool ws_internal_renderer_progress(const ws_renderer* r, uint32_t* started_seq);
// a renderer that runs beside others (a slot of a view batch with several frames in flight): see ws_api.cpp `throughput_mode`
void ws_internal_renderer_set_throughput_mode(ws_renderer* r, bool on);

// opaque handle definitions -------------------------------------------------------------------------
// The depth sort of a frame (V keys + store index + footprint word):
//   DS_SCAN      four 8-bit passes of tile histograms -> column scan -> scatter: 12 launches (the default since round 1)
//   DS_ONESWEEP  fat-tile one-sweep: one histogram launch + one launch per pass (round 4; measured variant)
//   DS_COOP      the same kernels as ONE launch with device-wide barriers between the passes (round 4, measured variant;
//                one frame at a time only: several such launches in flight starve each other of workgroup slots)
// Inputs beyond the fat forms' capacity (FAT_MAX_GRID x 8192 pairs) take DS_SCAN.
enum DepthSortMode { DS_SCAN = 0, DS_ONESWEEP = 2, DS_COOP = 3 };

struct ws_context {
    int device = 0;
    hipDeviceProp_t props;
    int depth_sort_mode = 0;  // DepthSortMode, WS_DEPTH_SORT = scan | onesweep | coop
    int dsort_fat_grid = 0;   // WS_DSORT_FAT_GRID (tuning): workgroups of the fat-tile depth sort, 0 = automatic
    int blend_variant = 0;
    int debug_cut = 0;        // WS_DEBUG_CUT (analysis): 0 = whole frame
    int blend_tpw_log2 = -1;  // WS_BLEND_TPW_LOG2: tiles per blend workgroup = 2^n (tuning); -1 = automatic
    int blend_lds_pad_kb = 0; // WS_BLEND_LDS_PAD_KB (tuning): unused dynamic LDS per blend workgroup
    int footprint = 0;        // WS_FOOTPRINT=ellipse: FP_ELLIPSE (the default is FP_RECT_PACKED, FP_RECT_COUNT for wide viewports)
    int batch_k1 = 1;         // WS_BATCH_K1=n: a view batch runs K1 once for groups of n frames (1 = every frame its own K1)
    int batch_threads = -1;   // WS_BATCH_THREADS: a view batch enqueues every slot's frames from its own host thread; -1 = for
                              //   point clouds of at most 512 Ki Gaussians (where one thread's launch rate is the limit), 0 / 1 = never / always
    bool tile_sort_wide = false; // WS_TILE_SORT=wide: single-pass tile-id sort up to 2048 binning tiles (launch_tile_sort_wide)
    int bin_request = 1;      // WS_BIN_SHIFT=0 | auto (default) | 1: BinRequest for frames that can use coarse binning
    int blend_dma = 0;        // WS_BLEND_DMA: the blend stages Splat records with LDS-DMA (global_load_lds_dwordx4 / _dword)
    int blend_split = -1;     // WS_BLEND_SPLIT: 4x4 binning tiles composited by two 4x2 workgroups each; -1 = when tiles < 2 x CUs
    int num_cus = 256;
    int depth_skip_top = 1;   // WS_DEPTH_SKIP_TOP=0: the depth sort's last pass always runs (A/B)
    int blend_order = -1;     // WS_BLEND_ORDER: -1 automatic (longest list first unless the renderer is a slot of a batch with frames in flight), 0 image order, 1 longest first
    int use_graph = 0;        // WS_GRAPH=1: prepare() on a real stream replays a captured frame graph instead of enqueueing 22
                              //   launches (opt-in: on ROCm 7.2 legacy-NULL-stream work between two launches of a used
                              //   executable graph makes the next launch fault)
    uint32_t tile_qw = 4, tile_qh = 4;  // WS_TILE_SHAPE = 2x2 | 4x2 | 4x4 (default: 32x32-px binning tiles)
    int batch_queue_depth = -1;    // ws_context_config::batch_queue_depth: -1 = the view batch's default
    bool capture = false;          // ws_context_config::capture: renderers keep source indices / debug words (tests)
    bool render_views_fast_blend = false;
    bool ply_decode_host = false;
    int depth_digit_bits = 0;      // ws_context_config::depth_digit_bits: 0 = default, 8 / 9 force the depth sort's digit width
    int blend_async = -1;          // ws_context_config::blend_async: -1 = default, 0 / 1 = k_blend / k_blend2 (barrier-free staging)
    int depth_tile_kpt = 0;        // ws_context_config::depth_tile_kpt (9-bit digits, A/B): 0 = by input size, 4 / 8 keys per thread
    // Is this context drawing on several streams in turn (a hand-rolled pipeline of renderers with frames in flight), or one
    // frame at a time?  The stream of the latest prepare() and how many consecutive prepare() calls used that same stream;
    // read by the automatic choice of the blend's workgroup order (ws_api.cpp).  Relaxed atomics: renderers of one context
    // may be driven from several host threads (the submission threads of a view batch).
    std::atomic<void*> last_prepare_stream{nullptr};
    std::atomic<uint32_t> same_stream_run{0};
};

struct ws_pointcloud {
    ws_context* ctx = nullptr;
    uint32_t num_points = 0;
    uint32_t sh_deg = 0;
    bool compressed = false;
    ws_aabb bbox{};
    float center[3] = {0, 0, 0};
    bool has_up = false;
    float up[3] = {0, 0, 0};
    bool has_mip = false, mip = false;
    bool has_kernel_size = false;
    float kernel_size = 0.f;
    bool has_background = false;
    float background[3] = {0, 0, 0};
    ws_gaussian_quantization quant{};
    // device memory
    uint4* planes = nullptr;         // uncompressed
    uint8_t* gaussians_c = nullptr;  // compressed
    uint8_t* sh_bytes = nullptr;
    uint8_t* covars = nullptr;
    size_t device_bytes = 0;
};
