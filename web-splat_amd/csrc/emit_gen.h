// emit_gen.h -- generation of the (tile id, splat) entries of one EMIT_TILE-entry slice, in draw order.
//
// Used by k_bin_emit (raster.hip); kept separate from the kernel because a sort pass can generate its own input
// with it (tried for the tile-id sort's first pass, not faster: see k_bin_emit).
//
// Every entry needs its owner: the draw position k with off[k] <= e < off[k] + cnt[k].  The owners of a whole slice
// are found at once: every owning position drops its index on its first entry (atomicMax, so that zero-footprint
// positions, which share an offset with their successor, lose), and an inclusive max-scan spreads it over the
// entries.  (A per-entry binary search over the offsets was 12 dependent, bank-conflicting LDS reads per entry.)
#pragma once

#include <hip/hip_runtime.h>

#include "footprint.h"
#include "ws_internal.h"

namespace ws {
namespace emit {

constexpr int THREADS = 256;
constexpr int EPT = EMIT_TILE / THREADS;  // entries per thread in the scan
constexpr int OFF_WORDS = EMIT_TILE + 2;        // s_off
constexpr int OWN_WORDS = EMIT_TILE + THREADS;  // s_own (bank-skewed)
__device__ __forceinline__ uint32_t pad(uint32_t i) { return i + i / EPT; }  // per-thread blocks, bank-skewed

struct Source {
    const uint32_t* sorted_idx;   // [V] store indices in draw order
    const uint32_t* fp_sorted;    // [V] footprint words by draw position (FP_RECT_PACKED: the packed rectangle)
    const uint32_t* sorted_idx_alt;  // both, where they are when the depth sort's last pass had nothing to do
    const uint32_t* fp_sorted_alt;   //   (FrameCounters::depth_skip_top; nullptr: that sort never skips)
    const uint8_t* splats;        // [V] x 20 B Splat records, by store index (the other footprint modes)
    const uint32_t* offsets;      // [V] exclusive prefix of tiles touched, by draw position
    const uint32_t* emit_start;   // draw position owning entry m * EMIT_TILE
    const FrameCounters* counters;
    uint32_t tiles_x;
    float vw, vh;                 // viewport in pixels (the f32 values K1 used)
    uint32_t tile_w_log2, tile_h_log2;
    bool ellipse;                 // FP_ELLIPSE: per-row spans; FP_RECT_COUNT: the whole rectangle
};

// words 0..2 of a Splat record (v1, v2, pos: the geometry; pointcloud.rs:352-358)
struct Geom {
    uint32_t w0, w1, w2;
};
__device__ __forceinline__ Geom load_geom(const Source& src, uint32_t store_idx) {
    const uint32_t* sp = reinterpret_cast<const uint32_t*>(src.splats + (size_t)store_idx * SPLAT_STRIDE);
    Geom g;
    g.w0 = sp[0];
    g.w1 = sp[1];
    g.w2 = sp[2];
    return g;
}

struct Slice {
    uint32_t e0, ne;        // first entry, number of entries
    uint32_t s_lo, ns;      // draw positions [s_lo, s_lo + ns) own them
    bool in_lds;            // the offsets fit the LDS window (block-uniform); else they are read from global memory
    const uint32_t* goff;   // offsets + s_lo
    // offset of the slice's position k (its first entry)
    __device__ __forceinline__ uint32_t off_at(const uint32_t* s_off, uint32_t k) const { return in_lds ? s_off[k] : goff[k]; }
};

// All THREADS threads of the workgroup call this (it contains barriers).  d = total entries, v = visible splats.
__device__ __forceinline__ Slice slice_setup(const Source& src, uint32_t slice, uint32_t d, uint32_t v, uint32_t* s_off,
                                             uint32_t* s_own, uint32_t* s_wmax) {
    const int tid = threadIdx.x;
    Slice sl;
    sl.e0 = slice * EMIT_TILE;
    const uint32_t e1 = (d - sl.e0) < (uint32_t)EMIT_TILE ? d : sl.e0 + EMIT_TILE;
    sl.ne = e1 - sl.e0;
    // draw positions [s_lo, s_hi] own the entries [e0, e1)
    sl.s_lo = src.emit_start[slice];
    const uint32_t s_hi = (e1 < d) ? src.emit_start[slice + 1] : (v - 1u);
    // Positions that own entries of this slice number at most EMIT_TILE + 1, but visible splats with an EMPTY
    // tile rectangle (centre inside the 1.2x cull bounds, footprint off screen) can sit in between in any
    // number.  Normally the offsets fit the LDS window; when they do not (slices of far, one-tile splats with empty ones
    // in between: found in round 3 -- with 64-px binning tiles the first slices of every frame took the per-entry binary
    // search on global memory that used to stand here, 12 dependent loads per entry, and k_bin_emit needed 24.6 us for
    // HALF the entries it writes in 14.4) the owners are marked from global memory and nothing else changes.
    sl.ns = s_hi - sl.s_lo + 1u;
    sl.in_lds = sl.ns <= (uint32_t)EMIT_TILE + 2u;
    sl.goff = src.offsets + sl.s_lo;
    {
        if (sl.in_lds)
            for (uint32_t k = tid; k < sl.ns; k += THREADS) s_off[k] = sl.goff[k];
        for (uint32_t i = tid; i < (uint32_t)OWN_WORDS; i += THREADS) s_own[i] = 0u;
        __syncthreads();
        for (uint32_t k = tid; k < sl.ns; k += THREADS) {
            const uint32_t o = sl.off_at(s_off, k);
            const uint32_t f = o > sl.e0 ? o - sl.e0 : 0u;  // first entry of position k inside the slice
            if (f < sl.ne) atomicMax(&s_own[pad(f)], k);
        }
        __syncthreads();
        // inclusive max-scan over the slice, EPT consecutive entries per thread
        uint32_t own[EPT];
        uint32_t run = 0u;
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            run = max(run, s_own[tid * (EPT + 1) + j]);
            own[j] = run;
        }
        const int lane = tid & 63, wave = tid >> 6;
        uint32_t incl = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o, 64);
            if (lane >= o) incl = max(incl, t);
        }
        if (lane == 63) s_wmax[wave] = incl;
        uint32_t prefix = __shfl_up(incl, 1, 64);
        if (lane == 0) prefix = 0u;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < THREADS / 64; ++w)
            if (w < wave) prefix = max(prefix, s_wmax[w]);
#pragma unroll
        for (int j = 0; j < EPT; ++j) s_own[tid * (EPT + 1) + j] = max(own[j], prefix);
    }
    __syncthreads();
    return sl;
}

// Tile id of the k-th tile of the splat's footprint (footprint.h): the same function of the same 12 bytes K1 counted.
__device__ __forceinline__ uint32_t tile_of(const Source& src, const Geom& g, uint32_t k) {
#ifdef WS_EXPERIMENTAL  // (binning by the kept ellipse, WS_FOOTPRINT=ellipse: measured variant; the product build bins by the rectangle)
    const bool ellipse = src.ellipse;
#else
    const bool ellipse = false;
#endif
    const fp::Tiles ft = fp::setup(g.w0, g.w1, g.w2, src.vw, src.vh, src.tile_w_log2, src.tile_h_log2, ellipse);
    return fp::tile_at(ft, k, src.tiles_x, src.tile_w_log2, src.tile_h_log2);
}

// FP_RECT_PACKED: tile id of the k-th tile (row-major inside the rectangle) of the packed rectangle r.
__device__ __forceinline__ uint32_t tile_of_rect(uint32_t r, uint32_t k, uint32_t tiles_x) {
    const uint32_t x0 = r & 0xFFu, y0 = (r >> 8) & 0xFFu;  // x0 | y0 << 8 | (w - 1) << 16 | (h - 1) << 24
    const uint32_t w = ((r >> 16) & 0xFFu) + 1u;
    // k / w without the integer-division sequence: k < 2^16 (a rectangle has at most 256 x 256 tiles), far inside the
    // range where the float path with one correction step is exact
    uint32_t q = (uint32_t)((float)k * __builtin_amdgcn_rcpf((float)w));
    uint32_t rem = k - q * w;
    if ((int32_t)rem < 0) { q -= 1u; rem += w; }
    if (rem >= w) { q += 1u; rem -= w; }
    return (y0 + q) * tiles_x + (x0 + rem);
}

// Entry el (0 <= el < sl.ne) of the slice: tile id and splat (store index).
template <bool PACKED>
__device__ __forceinline__ void entry(const Source& src, const Slice& sl, const uint32_t* s_off, const uint32_t* s_own,
                                      uint32_t el, uint32_t* key, uint32_t* val) {
    const uint32_t e = sl.e0 + el;
    const uint32_t lo = s_own[pad(el)];  // the LAST position whose offset is <= e (zero-footprint positions share their successor's)
    const uint32_t pos = sl.s_lo + lo;
    *val = src.sorted_idx[pos];
    const uint32_t k = e - sl.off_at(s_off, lo);
    *key = PACKED ? tile_of_rect(src.fp_sorted[pos], k, src.tiles_x) : tile_of(src, load_geom(src, *val), k);
}

}  // namespace emit
}  // namespace ws
