// preprocess.hip -- K1 / K1c for gfx950: per-Gaussian cull, EWA projection, SH colour, ordered compaction.
//
// Replaces src/shaders/preprocess.wgsl:163-280 (uncompressed) and
// src/shaders/preprocess_compressed.wgsl:206-332 (c3dgs) of the reference.
//
// MI355X design (not a translation of the WGSL):
//   * The point cloud lives in HBM as eight planes of 16-B chunks (ws_internal.h), so every load
//     instruction of a wave is one fully coalesced 1-KiB read, and a culled Gaussian costs 16 B
//     instead of 124 B: SH / covariance planes are only touched by lanes that survive the cull.
//   * Compaction is ORDERED (store order == Gaussian index order) through a ticketed, wave-parallel
//     decoupled look-back over 1024-Gaussian blocks: one 32-bit {flag,count} word per block, no fence
//     needed because the word is its own payload.  The reference's atomicAdd(keys_size) order is
//     nondeterministic (preprocess.wgsl:262); ordered compaction makes equal-depth ties, and hence the
//     image, reproducible across runs and ranks.
//   * Besides the reference's outputs (Splat 20 B, depth key) the kernel emits the number of binning tiles the
//     splat's kept ellipse reaches (4 B, footprint.h) for the binning stage that replaces the hardware rasteriser.
//
// This file is compiled with -ffp-contract=off: f32 operations happen in source order, which is the
// WGSL expression order, so results can be compared with the CPU restatement at the f16-ulp level.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "footprint.h"
#include "lookback.h"
#include "ws_internal.h"

namespace ws {

namespace {

#ifndef WS_K1_THREADS
#define WS_K1_THREADS 256
#endif
constexpr int K1_THREADS = WS_K1_THREADS;
#ifndef WS_K1_ITEMS
#define WS_K1_ITEMS 4
#endif
constexpr int K1_ITEMS = WS_K1_ITEMS;  // Gaussians per thread (measured on c2: 2 -> 59 us, 4 -> 56, 8 -> 70; 1 -> 80:
                                       // one ticket per 256 Gaussians makes the ~11 ns dispenser atomic the bottleneck)
#ifndef WS_K1_BACK_GROUP
#define WS_K1_BACK_GROUP 2
#endif
constexpr int K1_BACK_GROUP = WS_K1_BACK_GROUP;  // survivors whose covariance + SH planes are in flight together
// (1 instead of 2: 117 instead of 143 VGPRs, four waves per SIMD instead of three -- and the same 53 us: the kernel
//  runs its ~36 us of HBM traffic and ~20 us of VALU work one after the other, whatever the occupancy)

__device__ __forceinline__ float h2f(uint32_t h) { return __half2float(__ushort_as_half((unsigned short)(h & 0xFFFFu))); }
__device__ __forceinline__ uint32_t f2h(float f) { return (uint32_t)__half_as_ushort(__float2half_rn(f)); }

// Arithmetic policy.  Default (WS_K1_STRICT=1): correctly rounded f32 division / sqrt and source-order
// arithmetic everywhere, so that the visible set, the store order, the depth keys AND every f16 Splat field agree
// with the CPU restatement of preprocess.wgsl at the 1-ulp level.  -DWS_K1_STRICT=0 switches the BACK end
// (covariance, eigen-decomposition, SH -- everything that is rounded to f16 anyway) to the hardware's 1-ulp
// rcp / sqrt / rsq plus FMA contraction: measured 54 -> 48 us on 1.2 M Gaussians (the kernel is VALU-issue
// bound, profiles/r01), but the eigenvector of nearly isotropic splats is ill-conditioned and moves by many f16
// ulps, and a few image pixels leave the 2e-3 tolerance -- not worth 6 us, so it is off by default.
#ifndef WS_K1_STRICT
#define WS_K1_STRICT 1
#endif
__device__ __forceinline__ float qdiv(float a, float b) {
#if WS_K1_STRICT
    return a / b;
#else
    return a * __builtin_amdgcn_rcpf(b);
#endif
}
__device__ __forceinline__ float qsqrt(float a) {
#if WS_K1_STRICT
    return sqrtf(a);
#else
    return __builtin_amdgcn_sqrtf(a);
#endif
}

// SH basis constants: preprocess.wgsl:4-23
__device__ constexpr float SH_C0 = 0.28209479177387814f;
__device__ constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2_0 = 1.0925484305920792f, SH_C2_1 = -1.0925484305920792f,
                           SH_C2_2 = 0.31539156525252005f, SH_C2_3 = -1.0925484305920792f,
                           SH_C2_4 = 0.5462742152960396f;
__device__ constexpr float SH_C3_0 = -0.5900435899266435f, SH_C3_1 = 2.890611442640554f,
                           SH_C3_2 = -0.4570457994644658f, SH_C3_3 = 0.3731763325901154f,
                           SH_C3_4 = -0.4570457994644658f, SH_C3_5 = 1.445305721320277f,
                           SH_C3_6 = -0.5900435899266435f;

struct Sh16 {
    float c[16][3];
};

// preprocess.wgsl:124-154 evaluate_sh, all three channels at once.  WGSL evaluates `C * p * sh` left to right, so the
// channel-independent factor (C * p) of every term is computed once (bit-identical to the per-channel form); within a
// channel the terms of a band are summed left to right and then added to the result, as the shader does.
__device__ __forceinline__ void eval_sh3(const Sh16& sh, float x, float y, float z, uint32_t deg, float out[3]) {
    float bas[16];
    bas[0] = SH_C0;
    bas[1] = -SH_C1 * y;
    bas[2] = SH_C1 * z;
    bas[3] = SH_C1 * x;  // subtracted
    const float xx = x * x, yy = y * y, zz = z * z;
    const float xy = x * y, yz = y * z, xz = x * z;
    bas[4] = SH_C2_0 * xy;
    bas[5] = SH_C2_1 * yz;
    bas[6] = SH_C2_2 * (2.0f * zz - xx - yy);
    bas[7] = SH_C2_3 * xz;
    bas[8] = SH_C2_4 * (xx - yy);
    bas[9] = SH_C3_0 * y * (3.0f * xx - yy);
    bas[10] = SH_C3_1 * xy * z;
    bas[11] = SH_C3_2 * y * (4.0f * zz - xx - yy);
    bas[12] = SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
    bas[13] = SH_C3_4 * x * (4.0f * zz - xx - yy);
    bas[14] = SH_C3_5 * z * (xx - yy);
    bas[15] = SH_C3_6 * x * (xx - 3.0f * yy);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float result = bas[0] * sh.c[0][ch];
        if (deg > 0u) {
            result += bas[1] * sh.c[1][ch] + bas[2] * sh.c[2][ch] - bas[3] * sh.c[3][ch];
            if (deg > 1u) {
                result += bas[4] * sh.c[4][ch] + bas[5] * sh.c[5][ch] + bas[6] * sh.c[6][ch] + bas[7] * sh.c[7][ch] +
                          bas[8] * sh.c[8][ch];
                if (deg > 2u) {
                    result += bas[9] * sh.c[9][ch] + bas[10] * sh.c[10][ch] + bas[11] * sh.c[11][ch] +
                              bas[12] * sh.c[12][ch] + bas[13] * sh.c[13][ch] + bas[14] * sh.c[14][ch] +
                              bas[15] * sh.c[15][ch];
                }
            }
        }
        result += 0.5f;
        out[ch] = result;
    }
}

struct SplatOut {
    uint32_t w[5];  // Splat: v(4 x f16) pos(2 x f16) color(4 x f16)
    uint32_t key;
    uint32_t fp;    // the binning footprint word (ws_internal.h FootprintMode)
};

#define VM(c, r) (p.cam.view[(c)*4 + (r)])
#define PM(c, r) (p.cam.proj[(c)*4 + (r)])

// depth key: preprocess.wgsl:270-273 / preprocess_compressed.wgsl:322-325.  Always strict; znear / zfar are
// recovered from the projection matrix on the host with the same two f32 divisions the shader does.
template <bool COMPRESSED>
__device__ __forceinline__ uint32_t k1_depth_key(const K1Params& p, float clip_z) {
    if (!COMPRESSED) return __float_as_uint(p.zfar - clip_z);
    const float kf = 16777215.0f - (clip_z - p.znear) / (p.zfar - p.znear) * 16777215.0f;
    uint32_t k = 0u;  // WGSL u32(f32) saturates
    if (kf > 0.0f) k = (kf >= 4294967296.0f) ? 0xFFFFFFFFu : (uint32_t)kf;
    return k;
}

// From the frustum test onward: preprocess.wgsl:194-273 / preprocess_compressed.wgsl:234-325.
template <bool COMPRESSED, int FPMODE>
__device__ void k1_math(const K1Params& p, const float xyz[3], const float camspace[4], const float pos2d[4],
                        float opacity, const float cov6[6], const Sh16& sh, SplatOut* out) {
#if !WS_K1_STRICT
#pragma clang fp contract(fast)
#endif
    // fade-in (preprocess.wgsl:196-203).  p.fade_done (host, uniform): walltime is so far past the largest
    // possible dd = 5 * |centre - xyz| / scene_extend that smoothstep() is exactly 1 for every Gaussian.
    float scale_mod = 1.0f;
    if (!p.fade_done) {
        const float walltime = p.rs.walltime;
        scale_mod = 0.0f;
        const float ddx = p.rs.scene_center[0] - xyz[0], ddy = p.rs.scene_center[1] - xyz[1],
                    ddz = p.rs.scene_center[2] - xyz[2];
        const float dd = 5.0f * sqrtf(ddx * ddx + ddy * ddy + ddz * ddz) / p.rs.scene_extend;
        if (walltime > dd) {
            float t = (walltime - dd - 0.0f) / (1.0f - 0.0f);
            t = fminf(fmaxf(t, 0.0f), 1.0f);
            scale_mod = t * t * (3.0f - 2.0f * t);
        }
    }
    const float scaling = p.rs.gaussian_scaling * scale_mod;

    // Vrk = sym(cov6) * scaling * scaling
    float V[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) V[i] = cov6[i] * scaling * scaling;
    const float v00 = V[0], v01 = V[1], v02 = V[2], v11 = V[3], v12 = V[4], v22 = V[5];

    const float fx = p.cam.focal[0], fy = p.cam.focal[1];
    const float cz = camspace[2];
#if WS_K1_STRICT
    const float j00 = fx / cz;
    const float j02 = -(fx * camspace[0]) / (cz * cz);
    const float j11 = -fy / cz;
    const float j12 = (fy * camspace[1]) / (cz * cz);
#else
    const float rz = __builtin_amdgcn_rcpf(cz);
    const float j00 = fx * rz;
    const float j02 = -(fx * camspace[0]) * rz * rz;
    const float j11 = -fy * rz;
    const float j12 = (fy * camspace[1]) * rz * rz;
#endif
    // W = transpose(mat3(view)) -> column k of W is row k of the view rotation;
    // T = W * J: T[0] = W[0]*j00 + W[1]*0 + W[2]*j02 ; T[1] = W[0]*0 + W[1]*j11 + W[2]*j12 ; T[2] = 0.
    float t0[3], t1[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        // W[k][r] = view[r][k]  (view[c][r] is column c, row r)
        const float w0 = VM(r, 0), w1 = VM(r, 1), w2 = VM(r, 2);
        float s0 = w0 * j00;
        s0 += w1 * 0.0f;
        s0 += w2 * j02;
        t0[r] = s0;
        float s1 = w0 * 0.0f;
        s1 += w1 * j11;
        s1 += w2 * j12;
        t1[r] = s1;
    }
    // A = transpose(T) * Vrk : A[c][r] = sum_k Tt[k][r] * Vrk[c][k] = sum_k T[r][k] * Vrk[c][k]
    // rows r = 0,1 matter; Vrk symmetric, Vrk[c][k] = V(c,k)
    float a0[3], a1[3];  // a0[c] = A[c][0], a1[c] = A[c][1]
    {
        const float Vc[3][3] = {{v00, v01, v02}, {v01, v11, v12}, {v02, v12, v22}};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s = t0[0] * Vc[c][0];
            s += t0[1] * Vc[c][1];
            s += t0[2] * Vc[c][2];
            a0[c] = s;
            float u = t1[0] * Vc[c][0];
            u += t1[1] * Vc[c][1];
            u += t1[2] * Vc[c][2];
            a1[c] = u;
        }
    }
    // cov = A * T : cov[c][r] = sum_k A[k][r] * T[c][k]
    float cov00 = a0[0] * t0[0];
    cov00 += a0[1] * t0[1];
    cov00 += a0[2] * t0[2];
    float cov01 = a1[0] * t0[0];  // cov[0][1]: column 0, row 1
    cov01 += a1[1] * t0[1];
    cov01 += a1[2] * t0[2];
    float cov11 = a1[0] * t1[0];
    cov11 += a1[1] * t1[1];
    cov11 += a1[2] * t1[2];

    const float kernel_size = p.rs.kernel_size;
    if (p.rs.mip_splatting != 0u) {  // preprocess.wgsl:225-236
        const float det_0 = fmaxf(1e-6f, cov00 * cov11 - cov01 * cov01);
        const float det_1 = fmaxf(1e-6f, (cov00 + kernel_size) * (cov11 + kernel_size) - cov01 * cov01);
        float coef = qsqrt(qdiv(det_0, det_1 + 1e-6f) + 1e-6f);
        if (det_0 <= 1e-6f || det_1 <= 1e-6f) coef = 0.0f;
        opacity *= coef;
    }
    const float diagonal1 = cov00 + kernel_size;
    const float offDiagonal = cov01;
    const float diagonal2 = cov11 + kernel_size;
    const float mid = 0.5f * (diagonal1 + diagonal2);
    const float hx = (diagonal1 - diagonal2) / 2.0f;
    const float radius = qsqrt(hx * hx + offDiagonal * offDiagonal);
    float lambda1, lambda2;
    if (!COMPRESSED) {
        lambda1 = mid + radius;
        lambda2 = fmaxf(mid - radius, 0.1f);
    } else {
        lambda1 = mid + fmaxf(radius, 0.1f);
        lambda2 = mid - fmaxf(radius, 0.1f);
    }
    const float dvx = offDiagonal, dvy = lambda1 - diagonal1;
    float ex = 1.0f, ey = 0.0f;  // normalize((0,0)) is undefined in WGSL; defined as (1,0) here
#if WS_K1_STRICT
    const float dlen = sqrtf(dvx * dvx + dvy * dvy);
    if (dlen > 0.0f) {
        ex = dvx / dlen;
        ey = dvy / dlen;
    }
#else
    const float dlen2 = dvx * dvx + dvy * dvy;
    if (dlen2 > 0.0f) {
        const float inv_len = __builtin_amdgcn_rsqf(dlen2);
        ex = dvx * inv_len;
        ey = dvy * inv_len;
    }
#endif
    const float s1 = qsqrt(2.0f * lambda1), s2 = qsqrt(2.0f * lambda2);
    const float v1x = s1 * ex, v1y = s1 * ey;
    const float v2x = s2 * ey, v2y = s2 * (-ex);
    const float vcx = qdiv(pos2d[0], pos2d[3]), vcy = qdiv(pos2d[1], pos2d[3]);

    // colour: preprocess.wgsl:255-260
    const float dx = xyz[0] - p.cam.view_inv[12], dy = xyz[1] - p.cam.view_inv[13], dz = xyz[2] - p.cam.view_inv[14];
#if WS_K1_STRICT
    const float dl = sqrtf(dx * dx + dy * dy + dz * dz);
    const float dirx = dx / dl, diry = dy / dl, dirz = dz / dl;
#else
    const float inv_dl = __builtin_amdgcn_rsqf(dx * dx + dy * dy + dz * dz);
    const float dirx = dx * inv_dl, diry = dy * inv_dl, dirz = dz * inv_dl;
#endif
    float rgb[3];
    eval_sh3(sh, dirx, diry, dirz, p.rs.max_sh_deg, rgb);
    const float cr = fmaxf(0.0f, rgb[0]), cg = fmaxf(0.0f, rgb[1]), cb = fmaxf(0.0f, rgb[2]);

    const float vw = p.cam.viewport[0], vh = p.cam.viewport[1];
    const uint32_t h0 = f2h(qdiv(v1x, vw)), h1 = f2h(qdiv(v1y, vh)), h2 = f2h(qdiv(v2x, vw)), h3 = f2h(qdiv(v2y, vh));
    const uint32_t h4 = f2h(vcx), h5 = f2h(vcy);
    out->w[0] = h0 | (h1 << 16);
    out->w[1] = h2 | (h3 << 16);
    out->w[2] = h4 | (h5 << 16);
    out->w[3] = f2h(cr) | (f2h(cg) << 16);
    out->w[4] = f2h(cb) | (f2h(opacity) << 16);

    out->key = k1_depth_key<COMPRESSED>(p, pos2d[2]);

    // Binning footprint of the kept ellipse a <= 2*CUTOFF (gaussian.wgsl:40-64), derived from the f16-ROUNDED splat so
    // that binning and blending agree on coverage.  The packed bounding rectangle by default; the tile COUNT of the
    // rectangle (wide viewports) or of the ellipse itself (WS_FOOTPRINT=ellipse) through footprint.h, whose tiles
    // k_bin_emit re-derives from the same 12 bytes.
    if (FPMODE == FP_RECT_PACKED) {
        const float q1x = h2f(h0), q1y = h2f(h1), q2x = h2f(h2), q2y = h2f(h3);
        const float m00 = q1x * vw, m01 = q2x * vw;
        const float m10 = -q1y * vh, m11 = -q2y * vh;
        const float det = m00 * m11 - m01 * m10;
        const float cx = (h2f(h4) * 0.5f + 0.5f) * vw;
        const float cy = (0.5f - h2f(h5) * 0.5f) * vh;
        const float rad = 2.1697873f * 1.00001f;  // sqrt(2*CUTOFF), padded
        const float exx = rad * qsqrt(m00 * m00 + m01 * m01) + 1e-3f;
        const float eyy = rad * qsqrt(m10 * m10 + m11 * m11) + 1e-3f;
        uint32_t rect = RECT_EMPTY;
        const bool ok = (fabsf(det) > 0.0f) && (fabsf(det) < 3.0e38f) && (fabsf(cx) < 1.0e9f) && (fabsf(cy) < 1.0e9f) &&
                        (exx < 1.0e9f) && (eyy < 1.0e9f);
        if (ok) {
            // pixel (x, y) has its centre at (x + 0.5, y + 0.5)
            float x_lo = ceilf(cx - exx - 0.5f), x_hi = floorf(cx + exx - 0.5f);
            float y_lo = ceilf(cy - eyy - 0.5f), y_hi = floorf(cy + eyy - 0.5f);
            x_lo = fmaxf(x_lo, 0.0f);
            y_lo = fmaxf(y_lo, 0.0f);
            x_hi = fminf(x_hi, vw - 1.0f);
            y_hi = fminf(y_hi, vh - 1.0f);
            if (x_lo <= x_hi && y_lo <= y_hi) {
                const uint32_t tx0 = (uint32_t)x_lo >> p.tile_w_log2, tx1 = (uint32_t)x_hi >> p.tile_w_log2;
                const uint32_t ty0 = (uint32_t)y_lo >> p.tile_h_log2, ty1 = (uint32_t)y_hi >> p.tile_h_log2;
                rect = rect_pack(tx0, ty0, tx1, ty1);  // < 256 tiles per axis (ws_renderer_prepare picks the mode by the viewport)
            }
        }
        out->fp = rect;
    } else {
        const fp::Tiles ft = fp::setup(out->w[0], out->w[1], out->w[2], vw, vh, p.tile_w_log2, p.tile_h_log2, FPMODE == FP_ELLIPSE);
        out->fp = fp::count(ft, p.tile_w_log2, p.tile_h_log2);
    }
}

// Per-Gaussian front end: load position (+ the few words needed later), clip, project, frustum-cull.
struct Front {
    float xyz[3];
    uint32_t w3;            // uncompressed: opacity f16 | pad ; compressed: opacity i8 | scale i8 | pad
    uint32_t geometry_idx;  // compressed only
    uint32_t sh_idx;        // compressed only
};

template <bool COMPRESSED>
__device__ __forceinline__ void k1_load_front(const K1Buffers& b, uint32_t idx, Front* f) {
    if (!COMPRESSED) {
        const uint4 c0 = b.planes[idx];
        f->xyz[0] = __uint_as_float(c0.x);
        f->xyz[1] = __uint_as_float(c0.y);
        f->xyz[2] = __uint_as_float(c0.z);
        f->w3 = c0.w;
        f->geometry_idx = f->sh_idx = 0u;
    } else {
        // GaussianCompressed, 24 B (pointcloud.rs:14-22): three 8-B loads per lane
        const uint2* g = reinterpret_cast<const uint2*>(b.gaussians_c + (size_t)idx * 24);
        const uint2 a = g[0], bb = g[1], cc = g[2];
        f->xyz[0] = __uint_as_float(a.x);
        f->xyz[1] = __uint_as_float(a.y);
        f->xyz[2] = __uint_as_float(bb.x);
        f->w3 = bb.y;
        f->geometry_idx = cc.x;
        f->sh_idx = cc.y;
    }
}

// view / projection of one position; returns "survives clip box and frustum"
template <bool COMPRESSED>
__device__ __forceinline__ bool k1_project(const K1Params& p, const float xyz[3], float camspace[4], float pos2d[4]) {
    // world-space clip box (preprocess.wgsl:177-179)
    if (xyz[0] < p.rs.clip_min[0] || xyz[1] < p.rs.clip_min[1] || xyz[2] < p.rs.clip_min[2] ||
        xyz[0] > p.rs.clip_max[0] || xyz[1] > p.rs.clip_max[1] || xyz[2] > p.rs.clip_max[2])
        return false;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float s = VM(0, r) * xyz[0];
        s += VM(1, r) * xyz[1];
        s += VM(2, r) * xyz[2];
        s += VM(3, r) * 1.0f;
        camspace[r] = s;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float s = PM(0, r) * camspace[0];
        s += PM(1, r) * camspace[1];
        s += PM(2, r) * camspace[2];
        s += PM(3, r) * camspace[3];
        pos2d[r] = s;
    }
    const float bounds = 1.2f * pos2d[3];
    const float z = pos2d[2] / pos2d[3];
    bool culled;
    if (!COMPRESSED)  // preprocess.wgsl:190-192
        culled = z <= 0.0f || z >= 1.0f || pos2d[0] < -bounds || pos2d[0] > bounds || pos2d[1] < -bounds ||
                 pos2d[1] > bounds;
    else  // preprocess_compressed.wgsl:231
        culled = z < 0.0f || z > 1.0f || pos2d[0] < -bounds || pos2d[0] > bounds || pos2d[1] < -bounds ||
                 pos2d[1] > bounds;
    return !culled;
}

// Raw back-end words of one uncompressed Gaussian: covariance plane + up to six SH planes.
struct RawBack {
    uint4 c1;
    uint4 sh[6];
};

// Issue the back-end loads of one Gaussian.  UNCONDITIONAL on purpose: a load inside a divergent branch makes
// the compiler drain vmcnt at the join, which serialises the items of a thread on HBM latency.  Lanes that were
// culled read the record of `safe_idx` (a line the wave has already touched) and ignore the result.
__device__ __forceinline__ void k1_back_load(const K1Params& p, const K1Buffers& b, uint32_t idx, RawBack* rb) {
    const uint32_t n = p.num_points;
    rb->c1 = b.planes[(size_t)1 * n + idx];
    // SH planes 2..7: 48 halves, element e = 3*coef + channel.  Only the planes the active degree needs are
    // fetched: deg0 -> 1 plane, deg1 -> 2, deg2 -> 4, deg3 -> 6 (uniform branch).
    const uint32_t deg = p.rs.max_sh_deg;
    const int nplanes = deg == 0u ? 1 : (deg == 1u ? 2 : (deg == 2u ? 4 : 6));
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        rb->sh[q] = make_uint4(0u, 0u, 0u, 0u);
        if (q < nplanes) rb->sh[q] = b.planes[(size_t)(2 + q) * n + idx];
    }
}

template <int FPMODE>
__device__ __forceinline__ void k1_back_math(const K1Params& p, const Front& f, const RawBack& rb, SplatOut* so) {
    float camspace[4], pos2d[4];
    (void)k1_project<false>(p, f.xyz, camspace, pos2d);  // same instruction sequence as the front end
    float cov6[6];
    cov6[0] = h2f(rb.c1.x);
    cov6[1] = h2f(rb.c1.x >> 16);
    cov6[2] = h2f(rb.c1.y);
    cov6[3] = h2f(rb.c1.y >> 16);
    cov6[4] = h2f(rb.c1.z);
    cov6[5] = h2f(rb.c1.z >> 16);
    uint32_t hw[24];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        hw[q * 4 + 0] = rb.sh[q].x;
        hw[q * 4 + 1] = rb.sh[q].y;
        hw[q * 4 + 2] = rb.sh[q].z;
        hw[q * 4 + 3] = rb.sh[q].w;
    }
    Sh16 sh;
#pragma unroll
    for (int e = 0; e < 48; ++e) sh.c[e / 3][e % 3] = h2f(hw[e / 2] >> ((e & 1) * 16));
    k1_math<false, FPMODE>(p, f.xyz, camspace, pos2d, h2f(f.w3), cov6, sh, so);
}

// Back end for one survivor of the COMPRESSED layout: de-quantise, gather the codebooks, run the maths.
template <int FPMODE>
__device__ __forceinline__ void k1_back_compressed(const K1Params& p, const K1Buffers& b, const Front& f, SplatOut* so) {
    float camspace[4], pos2d[4];
    (void)k1_project<true>(p, f.xyz, camspace, pos2d);
    float cov6[6];
    Sh16 sh;
#pragma unroll
    for (int c = 0; c < 16; ++c) sh.c[c][0] = sh.c[c][1] = sh.c[c][2] = 0.0f;
    // preprocess_compressed.wgsl:236-242
    const int op_i8 = (int)(signed char)(f.w3 & 0xFFu);
    const int sc_i8 = (int)(signed char)((f.w3 >> 8) & 0xFFu);
    const float opacity = ((float)op_i8 - (float)p.quant.opacity.zero_point) * p.quant.opacity.scale;
    const float scaling_factor =
        expf(((float)sc_i8 - (float)p.quant.scaling_factor.zero_point) * p.quant.scaling_factor.scale);
    const float s2 = scaling_factor * scaling_factor;
    const uint32_t* cv = reinterpret_cast<const uint32_t*>(b.covars + (size_t)f.geometry_idx * 12);
    const uint32_t c0 = cv[0], c1 = cv[1], c2 = cv[2];
    cov6[0] = h2f(c0) * s2;
    cov6[1] = h2f(c0 >> 16) * s2;
    cov6[2] = h2f(c1) * s2;
    cov6[3] = h2f(c1 >> 16) * s2;
    cov6[4] = h2f(c2) * s2;
    cov6[5] = h2f(c2 >> 16) * s2;
    // int8 SH record: 3*ncoef bytes, packed back to back, NOT 4-aligned in general
    // (preprocess_compressed.wgsl:147-171).  unpack4x8snorm(x) * 127 = max(x / 127, -1) * 127 maps -128 to -127 and,
    // in f32, every other int8 value EXACTLY to itself (checked for all 256 values, tests/test_oracle.py), so the
    // 48 divisions per Gaussian of the literal form are replaced by a clamp: bit-identical, ~500 VALU instructions
    // per survivor cheaper.
    const uint32_t ncoef = p.sh_deg_layout;
    uint32_t use = (p.rs.max_sh_deg + 1u) * (p.rs.max_sh_deg + 1u);
    if (use > ncoef) use = ncoef;
    const signed char* rec = reinterpret_cast<const signed char*>(b.sh_bytes) + (size_t)3 * ((size_t)f.sh_idx * ncoef);
    const float zp_dc = (float)p.quant.color_dc.zero_point, zp_rest = (float)p.quant.color_rest.zero_point;
    const float sc_dc = p.quant.color_dc.scale, sc_rest = p.quant.color_rest.scale;
    if (ncoef == 16u) {
        // degree-3 layout: 48-B records, 16-B aligned (the blob is 256-B aligned) -> three 16-B loads
        const uint4* r4 = reinterpret_cast<const uint4*>(rec);
        const uint4 q0 = r4[0], q1 = r4[1], q2 = r4[2];
        const uint32_t w[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if ((uint32_t)c < use) {
                const float zp = c == 0 ? zp_dc : zp_rest;
                const float sc = c == 0 ? sc_dc : sc_rest;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int e = c * 3 + j;
                    const int v = (int)(signed char)((w[e >> 2] >> ((e & 3) * 8)) & 0xFFu);
                    sh.c[c][j] = ((float)(v < -127 ? -127 : v) - zp) * sc;
                }
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if ((uint32_t)c < use) {
                const float zp = c == 0 ? zp_dc : zp_rest;
                const float sc = c == 0 ? sc_dc : sc_rest;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int v = (int)rec[c * 3 + j];
                    sh.c[c][j] = ((float)(v < -127 ? -127 : v) - zp) * sc;
                }
            }
        }
    }
    k1_math<true, FPMODE>(p, f.xyz, camspace, pos2d, opacity, cov6, sh, so);
}

// One workgroup = one ticket = K1_ITEMS x 256 consecutive Gaussians (1024): a single device-wide atomic per
// 1024 Gaussians keeps the ticket dispenser (~11 ns per returning atomic on one address) well below the
// kernel's HBM time.  Store order inside the block is (item, thread) = Gaussian index order.
#ifndef WS_K1_MINWAVES
#define WS_K1_MINWAVES 1
#endif
template <bool COMPRESSED, int FPMODE>
__global__ __launch_bounds__(K1_THREADS, WS_K1_MINWAVES) void k_preprocess(const K1Params p, const K1Buffers b) {
    WS_SETPRIO_K1();
    ws_trace_begin(b.trace);
    __shared__ uint32_t s_bid;
    __shared__ uint32_t s_cnt[K1_ITEMS][K1_THREADS / 64];
    __shared__ uint32_t s_base;
    __shared__ uint32_t s_t32[K1_THREADS / 64], s_t64[K1_THREADS / 64];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // Workgroup ids are handed out by an atomic ticket = START order: a workgroup only ever waits for workgroups that
    // already hold their slot.  (blockIdx order -- -DWS_BLOCKIDX_ORDER -- saves the ~11 ns the returning atomic costs
    // per workgroup, in series, but is NOT safe with several look-back kernels in flight: measured on MI355X with four
    // frames on four streams, spinners of one kernel held the slots the missing predecessor of another needed and
    // vice versa -- 6 ms and 38 ms per frame instead of 0.18 and 0.35, lookback.h.)
#ifdef WS_BLOCKIDX_ORDER
    if (tid == 0) s_bid = blockIdx.x;
#else
    if (tid == 0) s_bid = atomicAdd(&b.counters->k1_ticket, 1u);
#endif
    __syncthreads();
    const uint32_t bid = s_bid;
    const uint32_t n = p.num_points;
    const uint32_t block_base = bid * (K1_THREADS * K1_ITEMS);

    // ---- front end for all items: issue every position load first, then cull ---------------------------
    Front fr[K1_ITEMS];
    bool vis[K1_ITEMS];
    uint32_t lane_rank[K1_ITEMS];
#pragma unroll
    for (int it = 0; it < K1_ITEMS; ++it) {
        // unconditional (index clamped): four position loads in flight per lane, no vmcnt drain at a branch join
        const uint32_t idx = block_base + it * K1_THREADS + tid;
        k1_load_front<COMPRESSED>(b, idx < n ? idx : n - 1u, &fr[it]);
    }
#pragma unroll
    for (int it = 0; it < K1_ITEMS; ++it) {
        const uint32_t idx = block_base + it * K1_THREADS + tid;
        float camspace[4], pos2d[4];
        vis[it] = (idx < n) && k1_project<COMPRESSED>(p, fr[it].xyz, camspace, pos2d);
        const unsigned long long vmask = __ballot(vis[it]);
        lane_rank[it] = __builtin_amdgcn_mbcnt_hi((uint32_t)(vmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vmask, 0u));
        if (lane == 0) s_cnt[it][wave] = (uint32_t)__popcll(vmask);
    }
    __syncthreads();

    // ---- ordered compaction, part 1: block aggregate, published before the heavy part -------------------
    uint32_t off[K1_ITEMS];
    uint32_t block_cnt = 0;
#pragma unroll
    for (int it = 0; it < K1_ITEMS; ++it) {
#pragma unroll
        for (int w = 0; w < K1_THREADS / 64; ++w) {
            const uint32_t c = s_cnt[it][w];
            if (w == wave) off[it] = block_cnt;
            block_cnt += c;
        }
    }
    if (tid == 0) lb::st(b.block_status + bid, lb::pack(p.epoch, bid == 0 ? lb::FLAG_INCL : lb::FLAG_AGG, block_cnt));
    if (tid == 0 && bid == 0) {  // for the later kernels of the frame (k_bin_prefix, k_bin_emit, the blend)
        b.counters->epoch = p.epoch;
        b.counters->bin_request = FPMODE == FP_RECT_PACKED ? p.bin_request : (uint32_t)BIN_NEVER;
    }

    // ---- back end, only for survivors --------------------------------------------------------------------
    SplatOut so[K1_ITEMS];
#pragma unroll
    for (int it = 0; it < K1_ITEMS; ++it) {
        so[it].w[0] = so[it].w[1] = so[it].w[2] = so[it].w[3] = so[it].w[4] = 0u;
        so[it].key = 0u;
        so[it].fp = FPMODE == FP_RECT_PACKED ? RECT_EMPTY : 0u;
    }
    if (!COMPRESSED) {
        const uint32_t safe_idx = block_base < n ? block_base : 0u;  // culled lanes re-read this (cached) record
#pragma unroll
        for (int grp = 0; grp < K1_ITEMS; grp += K1_BACK_GROUP) {
            RawBack rb[K1_BACK_GROUP];
#pragma unroll
            for (int u = 0; u < K1_BACK_GROUP; ++u) {
                const int it = grp + u;
                k1_back_load(p, b, vis[it] ? block_base + it * K1_THREADS + tid : safe_idx, &rb[u]);
            }
#pragma unroll
            for (int u = 0; u < K1_BACK_GROUP; ++u) {
                const int it = grp + u;
                if (vis[it]) k1_back_math<FPMODE>(p, fr[it], rb[u], &so[it]);
            }
        }
    } else {
#pragma unroll
        for (int it = 0; it < K1_ITEMS; ++it)
            if (vis[it]) k1_back_compressed<FPMODE>(p, b, fr[it], &so[it]);
    }

    // ---- ordered compaction, part 2: look-back (wave 0) and scatter ----------------------------------------
    if (wave == 0) {
        // (measured: replacing the ordered look-back by one unordered atomicAdd per block does not change this kernel's
        // time -- determinism is free here)
        const uint32_t excl = lb::wave_lookback(b.block_status, bid, p.epoch, lane, &b.counters->overflow, 2u);
        if (lane == 0) {
            s_base = excl;
            if (bid != 0) lb::st(b.block_status + bid, lb::pack(p.epoch, lb::FLAG_INCL, excl + block_cnt));
            if (bid == gridDim.x - 1) b.counters->num_visible = excl + block_cnt;
        }
    }
    __syncthreads();
    const uint32_t base = s_base;
#pragma unroll
    for (int it = 0; it < K1_ITEMS; ++it) {
        if (vis[it]) {
            const uint32_t slot = base + off[it] + lane_rank[it];
            uint32_t* sp = reinterpret_cast<uint32_t*>(b.splats + (size_t)slot * SPLAT_STRIDE);
            if (SPLAT_STRIDE == 32u) {  // the whole sector in two 16-B stores (no partial-sector write)
                reinterpret_cast<uint4*>(sp)[0] = make_uint4(so[it].w[0], so[it].w[1], so[it].w[2], so[it].w[3]);
                reinterpret_cast<uint4*>(sp)[1] = make_uint4(so[it].w[4], 0u, 0u, 0u);
            } else {
                sp[0] = so[it].w[0];
                sp[1] = so[it].w[1];
                sp[2] = so[it].w[2];
                sp[3] = so[it].w[3];
                sp[4] = so[it].w[4];
            }
            b.keys[slot] = so[it].key;
            b.footprints[slot] = so[it].fp;
            if (b.src_index) b.src_index[slot] = block_base + it * K1_THREADS + tid;
        }
    }
    // ---- the frame's footprint totals at both binning granularities (ws_internal.h bin_shift_decide) -----------------
    if (FPMODE == FP_RECT_PACKED && p.bin_request == BIN_AUTO) {  // (uniform: only frames that let the device decide)
        uint32_t t32 = 0u, t64 = 0u;
#pragma unroll
        for (int it = 0; it < K1_ITEMS; ++it)
            if (vis[it]) {
                t32 += rect_tiles(so[it].fp);
                t64 += rect_tiles64(so[it].fp);
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            t32 += (uint32_t)__shfl_xor((int)t32, o, 64);
            t64 += (uint32_t)__shfl_xor((int)t64, o, 64);
        }
        if (lane == 0) {
            s_t32[wave] = t32;
            s_t64[wave] = t64;
        }
    }
    if (FPMODE == FP_RECT_PACKED && p.bin_request == BIN_AUTO) {  // (here, behind the stores: the kernel's registers are free)
        __syncthreads();
        if (tid == 0) {
            uint32_t t32 = 0u, t64 = 0u;
#pragma unroll
            for (int w = 0; w < K1_THREADS / 64; ++w) {
                t32 += s_t32[w];
                t64 += s_t64[w];
            }
            if (t32) {  // (returnless; one pair per workgroup that has a visible splat with a footprint)
                uint32_t* ts = b.counters->tile_sums + (bid & (TILE_SUM_SLOTS - 1)) * TILE_SUM_STRIDE;
                atomicAdd(ts, t32);
                atomicAdd(ts + 1, t64);
            }
        }
    }
    ws_trace_end(b.trace);
}

}  // namespace

#ifdef WS_EXPERIMENTAL  // measured-and-lost variants: compiled by `make experimental` only
#include "experimental/preprocess_multi.hip"
#endif

namespace {
typedef void (*K1Kernel)(const K1Params, const K1Buffers);
K1Kernel k1_kernel(bool compressed, int mode) {
#ifdef WS_EXPERIMENTAL  // (binning by the ellipse: measured variant)
    if (mode == FP_ELLIPSE) return compressed ? &k_preprocess<true, FP_ELLIPSE> : &k_preprocess<false, FP_ELLIPSE>;
#endif
    if (compressed) return mode == FP_RECT_COUNT ? &k_preprocess<true, FP_RECT_COUNT> : &k_preprocess<true, FP_RECT_PACKED>;
    return mode == FP_RECT_COUNT ? &k_preprocess<false, FP_RECT_COUNT> : &k_preprocess<false, FP_RECT_PACKED>;
}
}  // namespace

const void* preprocess_kernel_func(bool compressed, int footprint_mode) {
    return reinterpret_cast<const void*>(k1_kernel(compressed, footprint_mode));
}

uint32_t preprocess_blocks(uint32_t n) { return (n + K1_THREADS * K1_ITEMS - 1) / (K1_THREADS * K1_ITEMS); }

int launch_preprocess(const K1Params& p, const K1Buffers& b, bool compressed, int footprint_mode, hipStream_t stream) {
    const uint32_t blocks = preprocess_blocks(p.num_points);
    if (blocks == 0) return WS_OK;
    hipLaunchKernelGGL(k1_kernel(compressed, footprint_mode), dim3(blocks), dim3(K1_THREADS), 0, stream, p, b);
    WS_HIP(hipGetLastError());
    return WS_OK;
}

#ifdef WS_EXPERIMENTAL  // measured-and-lost variants: compiled by `make experimental` only
#include "experimental/preprocess_multi_host.hip"
#else
int launch_preprocess_multi(const K1Params*, const K1Buffers*, uint32_t, bool, int, hipStream_t) {
    return fail(WS_ERR_UNSUPPORTED, "one K1 launch per group of views is only in the experimental build");
}
#endif

}  // namespace ws
