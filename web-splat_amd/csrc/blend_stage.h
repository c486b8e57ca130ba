// blend_stage.h -- decode of one (tile, splat) entry into the blend's staged record, and the coverage mask of
// the tile's 8x8-pixel quadrants.  Shared by k_blend (device) and ws_debug_stage_splat (host: the CPU unit test
// checks the mask against a brute-force walk over the pixel centres).
//
// Reference: the vertex stage of gaussian.wgsl:29-57 builds, per splat, the quad  centre + M * (+-cut, +-cut)  and
// the fragment stage (gaussian.wgsl:59-66) keeps a fragment when a = |screen_pos|^2 <= 2*CUTOFF, screen_pos =
// M^-1 (pixel - centre).  Here the same test is evaluated per pixel in the exp2 domain:
//   a' = |I' * pixel_local + c|^2,  I' = sqrt(log2 e) * M^-1,  c = -I' * centre_local,  kept when a' <= CUT_A2.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define WS_HD __host__ __device__ __forceinline__
#else
#define WS_HD inline
#endif

namespace ws {
namespace stage {

constexpr float LOG2E_F = 1.4426950408889634f;
constexpr float SQRT_LOG2E_F = 1.2011224087864498f;

WS_HD float fast_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
WS_HD float fast_sqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}
WS_HD float med3(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// f16 bits -> f32 (exact)
WS_HD float half_bits(uint32_t h) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __half2float(__ushort_as_half((unsigned short)(h & 0xFFFFu)));
#else
    const uint32_t s = (h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    uint32_t bits;
    if (e == 0) {
        if (m == 0) {
            bits = s;
        } else {  // subnormal: m * 2^-24
            const float v = (float)m * 5.9604644775390625e-8f;
            union { float f; uint32_t u; } c;
            c.f = v;
            bits = c.u | s;
        }
    } else if (e == 31) {
        bits = s | 0x7F800000u | (m << 13);
    } else {
        bits = s | ((e + 112u) << 23) | (m << 13);
    }
    union { uint32_t u; float f; } c;
    c.u = bits;
    return c.f;
#endif
}

struct Staged {
    float i00, i01, c0;  // row 0 of the affine map (tile-local pixel -> sqrt(log2 e) * screen_pos)
    float i10, i11, c1;  // row 1
    float alpha, r, g, b;
    uint32_t mask;       // bit (qy * QW + qx): the kept ellipse may reach quadrant (qx, qy) of the tile
};

// Coverage of the QW x QH quadrants (8x8 pixels each; pixel centres of quadrant column c span
// [8c + 0.5, 8c + 7.5] in tile-local coordinates) by the kept ellipse
//   { d : A dx^2 + B2 dx dy + C dy^2 <= cut },  d = pixel - (cxl, cyl),   D = A C - B2^2 / 4 > 0.
// Per band of quadrants (y in [y0, y1] around the centre) the ellipse's x-range over the band is exact: the right
// boundary x_r(y) = k y + sqrt(cut / A - (D / A^2) y^2) is concave, so its maximum over the band is taken at the
// ordinate of the ellipse's rightmost point clamped into the band (and symmetrically on the left); the intersection
// of a convex set with a vertical slab is non-empty iff the x-projections overlap.  Every quantity is padded by
// the rounding it can carry, towards "covered": the mask only prunes work, the per-pixel test decides.
template <int QW, int QH>
WS_HD uint32_t quadrant_mask(float A, float B2, float C, float D, float cxl, float cyl, float cut) {
    const float cutp = cut * 1.0001f + 1e-4f;
    const float invA = fast_rcp(A), invC = fast_rcp(C), invD = fast_rcp(D);
    const float ymax = fast_sqrt(cutp * A * invD) * 1.00001f + 1e-3f;  // half height of the ellipse
    const float xmax = fast_sqrt(cutp * C * invD);                     // half width
    const float k = -0.5f * B2 * invA;          // centre line of the horizontal chords: x = k y
    const float ys = -0.5f * B2 * invC * xmax;  // ordinate of the rightmost point (leftmost: -ys)
    const float cA = cutp * invA;
    const float dA2 = D * invA * invA;
    const float rpad = 8e-6f * cA;
    uint32_t mask = 0u;
#pragma unroll
    for (int r = 0; r < QH; ++r) {
        const float y0 = (float)(8 * r) + 0.5f - cyl, y1 = y0 + 7.0f;
        const float lo = fmaxf(y0, -ymax), hi = fminf(y1, ymax);
        if (lo <= hi) {
            const float yr = med3(ys, lo, hi), yl = med3(-ys, lo, hi);
            const float sr = fast_sqrt(fmaxf(cA - dA2 * yr * yr, 0.0f) + rpad);
            const float sl = fast_sqrt(fmaxf(cA - dA2 * yl * yl, 0.0f) + rpad);
            const float kr = k * yr, kl = k * yl;
            const float x1 = cxl + (kr + sr) + (4e-6f * (fabsf(kr) + sr + fabsf(cxl)) + 2e-3f);
            const float x0 = cxl + (kl - sl) - (4e-6f * (fabsf(kl) + sl + fabsf(cxl)) + 2e-3f);
            // columns c with 8c + 0.5 <= x1 and 8c + 7.5 >= x0
            const float fhi = med3(floorf((x1 - 0.5f) * 0.125f), -1.0f, (float)(QW - 1));
            const float flo = med3(ceilf((x0 - 7.5f) * 0.125f), 0.0f, (float)QW);
            if (flo <= fhi) {
                const uint32_t clo = (uint32_t)flo, chi = (uint32_t)fhi;
                mask |= ((2u << chi) - (1u << clo)) << (r * QW);
            }
        }
    }
    return mask;
}

// One entry: the 20-B Splat record (pointcloud.rs:352-358: v1 f16x2, v2 f16x2, pos f16x2, colour f16x4) ->
// staged record + quadrant mask, for the tile whose top-left pixel is (tile_x0, tile_y0).
template <int QW, int QH>
WS_HD Staged decode(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4, float W, float H, float tile_x0,
                    float tile_y0, float cut) {
    const float v1x = half_bits(w0), v1y = half_bits(w0 >> 16), v2x = half_bits(w1), v2y = half_bits(w1 >> 16);
    const float m00 = v1x * W, m01 = v2x * W;
    const float m10 = -v1y * H, m11 = -v2y * H;
    const float det = m00 * m11 - m01 * m10;
    const float inv = SQRT_LOG2E_F / det;
    const float cxl = (half_bits(w2) * 0.5f + 0.5f) * W - tile_x0;  // centre, tile-local pixels
    const float cyl = (0.5f - half_bits(w2 >> 16) * 0.5f) * H - tile_y0;
    Staged s;
    s.i00 = m11 * inv;
    s.i01 = -m01 * inv;
    s.i10 = -m10 * inv;
    s.i11 = m00 * inv;
    s.c0 = -(s.i00 * cxl + s.i01 * cyl);
    s.c1 = -(s.i10 * cxl + s.i11 * cyl);
    // a'(d) = A dx^2 + B2 dx dy + C dy^2 around the centre;  A C - B2^2/4 = det(I')^2 = (log2 e / det M)^2 (no
    // cancellation, unlike the difference of the products)
    const float A = s.i00 * s.i00 + s.i10 * s.i10, C = s.i01 * s.i01 + s.i11 * s.i11;
    const float B2 = 2.0f * (s.i00 * s.i01 + s.i10 * s.i11);
    const float dI = inv * SQRT_LOG2E_F;
    s.mask = quadrant_mask<QW, QH>(A, B2, C, dI * dI, cxl, cyl, cut);
    s.alpha = half_bits(w4 >> 16);
    s.r = half_bits(w3);
    s.g = half_bits(w3 >> 16);
    s.b = half_bits(w4);
    return s;
}

}  // namespace stage
}  // namespace ws
