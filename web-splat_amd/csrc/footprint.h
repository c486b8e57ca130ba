// footprint.h -- the binning tiles a splat's kept ellipse reaches: per tile row one column span.
//
// The reference never generates work outside the splat's oriented quad (gaussian.wgsl:40-53: the quad is spanned by
// v1, v2) and discards every fragment outside the ellipse a <= 2*CUTOFF (gaussian.wgsl:61-64).  Binning by the ellipse's
// axis-aligned bounding rectangle (rounds 1-2) listed every tile of the rectangle: 18 % more (tile, splat) entries than
// the ellipse touches on the 1 M / 1080p scene (scripts/footprint_study.py), each of them written, sorted twice and
// mostly staged.  Here a splat lists exactly the tiles whose pixel-centre box the kept ellipse can reach:
//   rows    the tile rows of the bounding rectangle (K1's derivation, unchanged),
//   columns per row the exact x-range of the ellipse over the row's band of pixel centres (the band-exact construction of
//           blend_stage.h's quadrant mask, at binning-tile size), padded by its rounding towards "covered".
// The per-pixel test of the blend still decides what is drawn: a tile that is dropped here holds no pixel centre that
// passes it, so the image does not change.
//
// Used by K1 (the COUNT of tiles, which rides through the depth sort as the splat's companion value) and by k_bin_emit
// (the k-th tile of the footprint, re-derived from the 12 geometry bytes of the Splat record).  Both must agree exactly:
// every function here is compiled with FP contraction OFF whatever the translation unit's flags say, uses only
// operations whose result is a function of the operands alone (v_rcp_f32 / v_sqrt_f32 are), and takes its inputs from
// the f16-ROUNDED record, never from K1's unrounded intermediates.
// The host twin (ws_debug_footprint) feeds the CPU brute-force test.
#pragma once

#include <math.h>
#include <stdint.h>

#include "blend_stage.h"

namespace ws {
namespace fp {

struct Tiles {
    uint32_t tx0, ty0, tx1, ty1;  // bounding rectangle in binning tiles, inclusive, clipped to the viewport
    bool any;                     // the rectangle is non-empty
    bool exact;                   // the ellipse is well-conditioned: per-row spans; else every row spans the rectangle
    // the kept ellipse { d : A dx^2 + B2 dx dy + C dy^2 <= cut } in the exp2 domain of the blend (a' = log2e * a), around
    // the centre (cxl, cyl) given in pixels relative to the top-left pixel of tile (tx0, ty0)
    float cxl, cyl;
    float k;       // centre line of the horizontal chords: x = k y
    float ys;      // ordinate of the rightmost point (leftmost: -ys)
    float ymax;    // half height, padded
    float cA;      // cut / A
    float dA2;     // D / A^2
    float rpad;
};

// Splat words 0..2 (v1 f16x2, v2 f16x2, pos f16x2; pointcloud.rs:352-358), the viewport in pixels as the f32 the camera
// uniform holds, log2 of the binning tile size.
WS_HD Tiles setup(uint32_t w0, uint32_t w1, uint32_t w2, float vw, float vh, uint32_t tw_log2, uint32_t th_log2,
                  bool spans = true) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    Tiles t;
    t.tx0 = t.ty0 = t.tx1 = t.ty1 = 0u;
    t.any = false;
    t.exact = false;
    t.cxl = t.cyl = t.k = t.ys = t.ymax = t.cA = t.dA2 = t.rpad = 0.0f;
    const float q1x = stage::half_bits(w0), q1y = stage::half_bits(w0 >> 16);
    const float q2x = stage::half_bits(w1), q2y = stage::half_bits(w1 >> 16);
    // gaussian.wgsl:40-53: the quad's half axes in pixels (y flipped)
    const float m00 = q1x * vw, m01 = q2x * vw;
    const float m10 = -q1y * vh, m11 = -q2y * vh;
    const float det = m00 * m11 - m01 * m10;
    const float cx = (stage::half_bits(w2) * 0.5f + 0.5f) * vw;
    const float cy = (0.5f - stage::half_bits(w2 >> 16) * 0.5f) * vh;
    const float rad = 2.1697873f * 1.00001f;  // sqrt(2*CUTOFF), padded
    const float exx = rad * stage::fast_sqrt(m00 * m00 + m01 * m01) * 1.000001f + 1e-3f;
    const float eyy = rad * stage::fast_sqrt(m10 * m10 + m11 * m11) * 1.000001f + 1e-3f;
    const bool ok = (fabsf(det) > 0.0f) && (fabsf(det) < 3.0e38f) && (fabsf(cx) < 1.0e9f) && (fabsf(cy) < 1.0e9f) &&
                    (exx < 1.0e9f) && (eyy < 1.0e9f);
    if (!ok) return t;
    // pixel (x, y) has its centre at (x + 0.5, y + 0.5)
    float x_lo = ceilf(cx - exx - 0.5f), x_hi = floorf(cx + exx - 0.5f);
    float y_lo = ceilf(cy - eyy - 0.5f), y_hi = floorf(cy + eyy - 0.5f);
    x_lo = fmaxf(x_lo, 0.0f);
    y_lo = fmaxf(y_lo, 0.0f);
    x_hi = fminf(x_hi, vw - 1.0f);
    y_hi = fminf(y_hi, vh - 1.0f);
    if (!(x_lo <= x_hi && y_lo <= y_hi)) return t;
    t.tx0 = (uint32_t)x_lo >> tw_log2;
    t.tx1 = (uint32_t)x_hi >> tw_log2;
    t.ty0 = (uint32_t)y_lo >> th_log2;
    t.ty1 = (uint32_t)y_hi >> th_log2;
    t.any = true;
    if (!spans) return t;  // (FP_RECT_COUNT: every row spans the rectangle)
    // the blend's quadratic form (blend_stage.h decode): I' = sqrt(log2 e) * M^-1
    const float inv = stage::SQRT_LOG2E_F / det;
    const float i00 = m11 * inv, i01 = -m01 * inv, i10 = -m10 * inv, i11 = m00 * inv;
    const float A = i00 * i00 + i10 * i10, C = i01 * i01 + i11 * i11;
    const float B2 = 2.0f * (i00 * i01 + i10 * i11);
    const float dI = inv * stage::SQRT_LOG2E_F;
    const float D = dI * dI;  // A C - B2^2 / 4 = det(I')^2, without the cancellation
    // well-conditioned: everything the spans divide by is a normal, finite number
    t.exact = (A > 1e-30f) && (A < 1e30f) && (C > 1e-30f) && (C < 1e30f) && (D > 1e-30f) && (D < 1e30f);
    if (!t.exact) return t;
    const float cutp = (2.0f * 2.3539888583335364f * stage::LOG2E_F) * 1.0001f + 1e-4f;  // gaussian.wgsl:61, exp2 domain, padded
    const float invA = stage::fast_rcp(A), invC = stage::fast_rcp(C), invD = stage::fast_rcp(D);
    t.ymax = stage::fast_sqrt(cutp * A * invD) * 1.00001f + 1e-3f;
    const float xmax = stage::fast_sqrt(cutp * C * invD);
    t.k = -0.5f * B2 * invA;
    t.ys = -0.5f * B2 * invC * xmax;
    t.cA = cutp * invA;
    t.dA2 = D * invA * invA;
    t.rpad = 8e-6f * t.cA;
    t.cxl = cx - (float)(t.tx0 << tw_log2);
    t.cyl = cy - (float)(t.ty0 << th_log2);
    // (tile origins are exact in f32 up to 2^24 pixels; the centre keeps its few significant bits)
    t.exact = (t.ymax < 1e9f) && (xmax < 1e9f) && (fabsf(t.k) < 1e9f) && (t.dA2 < 1e30f) && (t.cA < 1e30f);
    return t;
}

// Tile row ty (ty0 <= ty <= ty1): the columns [*lo, *lo + count) the ellipse can reach, clipped to the rectangle.
// Pixel centres of the row span y in [ty * TH + 0.5, ty * TH + TH - 0.5]; of column c, x in [c * TW + 0.5, c * TW + TW - 0.5].
WS_HD uint32_t row_span(const Tiles& t, uint32_t ty, uint32_t tw_log2, uint32_t th_log2, uint32_t* lo) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const uint32_t w = t.tx1 - t.tx0 + 1u;
    *lo = t.tx0;
    if (!t.exact) return w;
    const float y0 = (float)((ty - t.ty0) << th_log2) + 0.5f - t.cyl;
    const float y1 = y0 + (float)((1u << th_log2) - 1u);
    const float blo = fmaxf(y0, -t.ymax), bhi = fminf(y1, t.ymax);
    if (!(blo <= bhi)) return 0u;
    // the right boundary x_r(y) = k y + sqrt(cut / A - (D / A^2) y^2) is concave: over the band its maximum sits at the
    // ordinate of the ellipse's rightmost point clamped into the band (symmetrically on the left)
    const float yr = stage::med3(t.ys, blo, bhi), yl = stage::med3(-t.ys, blo, bhi);
    const float sr = stage::fast_sqrt(fmaxf(t.cA - t.dA2 * yr * yr, 0.0f) + t.rpad);
    const float sl = stage::fast_sqrt(fmaxf(t.cA - t.dA2 * yl * yl, 0.0f) + t.rpad);
    const float kr = t.k * yr, kl = t.k * yl;
    const float x1 = t.cxl + (kr + sr) + (4e-6f * (fabsf(kr) + sr + fabsf(t.cxl)) + 2e-3f);
    const float x0 = t.cxl + (kl - sl) - (4e-6f * (fabsf(kl) + sl + fabsf(t.cxl)) + 2e-3f);
    // columns c (relative to tx0) with c * TW + 0.5 <= x1 and c * TW + TW - 0.5 >= x0
    const float tw = (float)(1u << tw_log2), inv_tw = 1.0f / tw;  // a power of two: exact
    const float fhi = stage::med3(floorf((x1 - 0.5f) * inv_tw), -1.0f, (float)(w - 1u));
    const float flo = stage::med3(ceilf((x0 - (tw - 0.5f)) * inv_tw), 0.0f, (float)w);
    if (!(flo <= fhi)) return 0u;
    *lo = t.tx0 + (uint32_t)flo;
    return (uint32_t)fhi - (uint32_t)flo + 1u;
}

// number of tiles of the footprint (what K1 stores per visible splat)
WS_HD uint32_t count(const Tiles& t, uint32_t tw_log2, uint32_t th_log2) {
    if (!t.any) return 0u;
    if (!t.exact) return (t.tx1 - t.tx0 + 1u) * (t.ty1 - t.ty0 + 1u);
    uint32_t n = 0u;
    for (uint32_t ty = t.ty0; ty <= t.ty1; ++ty) {
        uint32_t lo;
        n += row_span(t, ty, tw_log2, th_log2, &lo);
    }
    return n;
}

// tile id (ty * tiles_x + tx) of the k-th tile of the footprint, rows top to bottom, columns left to right
WS_HD uint32_t tile_at(const Tiles& t, uint32_t k, uint32_t tiles_x, uint32_t tw_log2, uint32_t th_log2) {
    uint32_t last = t.ty0 * tiles_x + t.tx0;
    for (uint32_t ty = t.ty0; ty <= t.ty1; ++ty) {
        uint32_t lo;
        const uint32_t c = row_span(t, ty, tw_log2, th_log2, &lo);
        if (k < c) return ty * tiles_x + lo + k;
        k -= c;
        if (c) last = ty * tiles_x + lo + c - 1u;
    }
    return last;  // k beyond the footprint: cannot happen when K1 and the caller agree (kept total for safety)
}

}  // namespace fp
}  // namespace ws
