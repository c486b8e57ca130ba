/*
 * ws_oracle.c -- CPU restatement of web-splat's render hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see ws_oracle.h).  PARITY STATUS (details in ws_oracle.h):
 * the sort is pinned by the reference's known-answer vector (gpu_rs.rs:295-331); K1, K1c
 * and the K6 fragment function are pinned by golden vectors produced by executing the
 * reference's WGSL source text (tests/golden/wgsl_*.npz); parity unpinned for the Rust
 * host math, the rasteriser interpolation / blend state and the loaders.
 *
 * Every function cites the reference file:line (relative to /root/reference) it
 * restates.  Arithmetic is f32 in the WGSL/Rust expression order; build with
 * -ffp-contract=off and without -ffast-math (see oracle/Makefile).
 *
 * Third-party arithmetic the reference pulls in and that is NOT vendored under
 * /root/reference (restated from the published algorithms):
 *   - cgmath (git ff840cbf): Matrix3::from(Quaternion), Quaternion::from(Matrix3),
 *     Matrix4::invert / inverse_transform, transpose, normalize.
 *   - half 2.6.0: f16::from_f32 (round-to-nearest-even), f16::to_f32.
 *   - naga/wgpu 25: WGSL builtins pack2x16float, unpack2x16float, unpack4x8snorm,
 *     normalize, length, distance, smoothstep, extractBits.
 */
#include "ws_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int wso_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* binary16                                                                   */
/* ------------------------------------------------------------------------- */

static inline uint32_t f32_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
static inline float bits_f32(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* half::f16::from_f32 -- IEEE round-to-nearest-even, overflow -> inf, NaN kept quiet */
uint16_t wso_f32_to_f16(float f) {
    uint32_t x = f32_bits(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t exp = (x >> 23) & 0xFFu;
    uint32_t man = x & 0x7FFFFFu;
    if (exp == 0xFFu) { /* inf / nan */
        if (man) return (uint16_t)(sign | 0x7E00u | (man >> 13));
        return (uint16_t)(sign | 0x7C00u);
    }
    int32_t e = (int32_t)exp - 127 + 15;
    if (e >= 0x1F) return (uint16_t)(sign | 0x7C00u); /* overflow */
    if (e <= 0) {                                     /* subnormal or zero */
        if (e < -10) return (uint16_t)sign;
        man |= 0x800000u;
        uint32_t shift = (uint32_t)(14 - e);
        uint32_t half_man = man >> shift;
        uint32_t rem = man & ((1u << shift) - 1u);
        uint32_t halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (half_man & 1u))) half_man++;
        return (uint16_t)(sign | half_man);
    }
    uint32_t half_man = man >> 13;
    uint32_t rem = man & 0x1FFFu;
    uint32_t out = sign | ((uint32_t)e << 10) | half_man;
    if (rem > 0x1000u || (rem == 0x1000u && (half_man & 1u))) out++; /* may carry into exponent: correct */
    return (uint16_t)out;
}

float wso_f16_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    if (exp == 0) {
        if (man == 0) return bits_f32(sign);
        /* subnormal: normalise */
        int e = -1;
        do {
            e++;
            man <<= 1;
        } while (!(man & 0x400u));
        man &= 0x3FFu;
        return bits_f32(sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13));
    }
    if (exp == 0x1F) return bits_f32(sign | 0x7F800000u | (man << 13));
    return bits_f32(sign | ((exp + 127 - 15) << 23) | (man << 13));
}

/* ------------------------------------------------------------------------- */
/* small matrix helpers (column-major, as cgmath / WGSL)                      */
/* ------------------------------------------------------------------------- */

#define M4(m, c, r) ((m)[(c) * 4 + (r)])
#define M3(m, c, r) ((m)[(c) * 3 + (r)])

static float det3(float a00, float a01, float a02, float a10, float a11, float a12, float a20, float a21,
                  float a22) {
    return a00 * (a11 * a22 - a12 * a21) - a01 * (a10 * a22 - a12 * a20) + a02 * (a10 * a21 - a11 * a20);
}

/* general 4x4 inverse by cofactors (cgmath Matrix4::invert; SquareMatrix::invert).
 * Returns 0 when singular. */
static int mat4_invert(const float* m, float* out) {
    float cof[16];
    for (int c = 0; c < 4; c++) {
        for (int r = 0; r < 4; r++) {
            /* minor of element (row r, col c) */
            float s[9];
            int k = 0;
            for (int cc = 0; cc < 4; cc++) {
                if (cc == c) continue;
                for (int rr = 0; rr < 4; rr++) {
                    if (rr == r) continue;
                    s[k++] = M4(m, cc, rr);
                }
            }
            float d = det3(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], s[8]);
            cof[c * 4 + r] = ((r + c) & 1) ? -d : d;
        }
    }
    float det = M4(m, 0, 0) * cof[0 * 4 + 0] + M4(m, 1, 0) * cof[1 * 4 + 0] + M4(m, 2, 0) * cof[2 * 4 + 0] +
                M4(m, 3, 0) * cof[3 * 4 + 0];
    if (det == 0.0f) return 0;
    float inv_det = 1.0f / det;
    /* inverse = adjugate / det; adjugate(row i, col j) = cofactor(row j, col i) */
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) M4(out, c, r) = cof[r * 4 + c] * inv_det;
    return 1;
}

static void mat4_mul(const float* a, const float* b, float* out) {
    float t[16];
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) {
            float s = M4(a, 0, r) * M4(b, c, 0);
            s += M4(a, 1, r) * M4(b, c, 1);
            s += M4(a, 2, r) * M4(b, c, 2);
            s += M4(a, 3, r) * M4(b, c, 3);
            t[c * 4 + r] = s;
        }
    memcpy(out, t, sizeof t);
}

static void mat4_transpose(const float* a, float* out) {
    float t[16];
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) t[c * 4 + r] = M4(a, r, c);
    memcpy(out, t, sizeof t);
}

/* cgmath: impl From<Quaternion<S>> for Matrix3<S>.  q = (s, x, y, z). */
void wso_quat_to_mat3(const float q[4], float m[9]) {
    float s = q[0], x = q[1], y = q[2], z = q[3];
    float x2 = x + x, y2 = y + y, z2 = z + z;
    float xx2 = x2 * x, xy2 = x2 * y, xz2 = x2 * z;
    float yy2 = y2 * y, yz2 = y2 * z, zz2 = z2 * z;
    float sy2 = y2 * s, sz2 = z2 * s, sx2 = x2 * s;
    /* Matrix3::new(c0r0, c0r1, c0r2, c1r0, ...) */
    m[0] = 1.0f - yy2 - zz2;
    m[1] = xy2 + sz2;
    m[2] = xz2 - sy2;
    m[3] = xy2 - sz2;
    m[4] = 1.0f - xx2 - zz2;
    m[5] = yz2 + sx2;
    m[6] = xz2 + sy2;
    m[7] = yz2 - sx2;
    m[8] = 1.0f - xx2 - yy2;
}

/* cgmath: impl From<Matrix3<S>> for Quaternion<S> (Shoemake) */
void wso_mat3_to_quat(const float m[9], float q[4]) {
    float trace = M3(m, 0, 0) + M3(m, 1, 1) + M3(m, 2, 2);
    float w, x, y, z;
    if (trace >= 0.0f) {
        float s = sqrtf(1.0f + trace);
        w = 0.5f * s;
        s = 0.5f / s;
        x = (M3(m, 1, 2) - M3(m, 2, 1)) * s;
        y = (M3(m, 2, 0) - M3(m, 0, 2)) * s;
        z = (M3(m, 0, 1) - M3(m, 1, 0)) * s;
    } else if (M3(m, 0, 0) > M3(m, 1, 1) && M3(m, 0, 0) > M3(m, 2, 2)) {
        float s = sqrtf((M3(m, 0, 0) - M3(m, 1, 1) - M3(m, 2, 2)) + 1.0f);
        x = 0.5f * s;
        s = 0.5f / s;
        y = (M3(m, 1, 0) + M3(m, 0, 1)) * s;
        z = (M3(m, 0, 2) + M3(m, 2, 0)) * s;
        w = (M3(m, 1, 2) - M3(m, 2, 1)) * s;
    } else if (M3(m, 1, 1) > M3(m, 2, 2)) {
        float s = sqrtf((M3(m, 1, 1) - M3(m, 0, 0) - M3(m, 2, 2)) + 1.0f);
        y = 0.5f * s;
        s = 0.5f / s;
        z = (M3(m, 2, 1) + M3(m, 1, 2)) * s;
        x = (M3(m, 1, 0) + M3(m, 0, 1)) * s;
        w = (M3(m, 2, 0) - M3(m, 0, 2)) * s;
    } else {
        float s = sqrtf((M3(m, 2, 2) - M3(m, 0, 0) - M3(m, 1, 1)) + 1.0f);
        z = 0.5f * s;
        s = 0.5f / s;
        x = (M3(m, 0, 2) + M3(m, 2, 0)) * s;
        y = (M3(m, 2, 1) + M3(m, 1, 2)) * s;
        w = (M3(m, 0, 1) - M3(m, 1, 0)) * s;
    }
    q[0] = w;
    q[1] = x;
    q[2] = y;
    q[3] = z;
}

/* src/camera.rs:207-214 world2view:
 *   rt = Matrix4::from(r); rt[0].w = t.x; rt[1].w = t.y; rt[2].w = t.z; rt[3].w = 1;
 *   rt.inverse_transform().unwrap().transpose()
 * (rt[c].w is column c, row 3: the translation sits in the bottom row.) */
void wso_world2view(const float r[9], const float t[3], float out[16]) {
    float rt[16];
    memset(rt, 0, sizeof rt);
    for (int c = 0; c < 3; c++)
        for (int rr = 0; rr < 3; rr++) M4(rt, c, rr) = M3(r, c, rr);
    M4(rt, 0, 3) = t[0];
    M4(rt, 1, 3) = t[1];
    M4(rt, 2, 3) = t[2];
    M4(rt, 3, 3) = 1.0f;
    float inv[16];
    if (!mat4_invert(rt, inv)) memset(inv, 0, sizeof inv);
    mat4_transpose(inv, out);
}

/* src/camera.rs:216-234 build_proj (fills p[col][row] then transposes) */
void wso_build_proj(float znear, float zfar, float fov_x, float fov_y, float out[16]) {
    float tan_half_fov_y = tanf(fov_y / 2.0f);
    float tan_half_fov_x = tanf(fov_x / 2.0f);
    float top = tan_half_fov_y * znear;
    float bottom = -top;
    float right = tan_half_fov_x * znear;
    float left = -right;
    float p[16];
    memset(p, 0, sizeof p);
    M4(p, 0, 0) = 2.0f * znear / (right - left);
    M4(p, 1, 1) = 2.0f * znear / (top - bottom);
    M4(p, 0, 2) = (right + left) / (right - left);
    M4(p, 1, 2) = (top + bottom) / (top - bottom);
    M4(p, 3, 2) = 1.0f;
    M4(p, 2, 2) = zfar / (zfar - znear);
    M4(p, 2, 3) = -(zfar * znear) / (zfar - znear);
    mat4_transpose(p, out);
}

float wso_fov2focal(float fov, float pixels) { return pixels / (2.0f * tanf(fov * 0.5f)); }
float wso_focal2fov(float focal, float pixels) { return 2.0f * atanf(pixels / (2.0f * focal)); }

/* pointcloud.rs:444-452: center = midpoint, radius = |max-min| / 2 */
float wso_aabb_radius(const wso_aabb* b) {
    float dx = b->min[0] - b->max[0], dy = b->min[1] - b->max[1], dz = b->min[2] - b->max[2];
    return sqrtf(dx * dx + dy * dy + dz * dz) / 2.0f;
}

/* src/camera.rs:26-35 */
void wso_fit_near_far(wso_camera* cam, const wso_aabb* bb) {
    float c[3];
    for (int i = 0; i < 3; i++) c[i] = bb->min[i] + (bb->max[i] - bb->min[i]) / 2.0f; /* cgmath EuclideanSpace::midpoint */
    float radius = wso_aabb_radius(bb);
    float dx = cam->position[0] - c[0], dy = cam->position[1] - c[1], dz = cam->position[2] - c[2];
    float distance = sqrtf(dx * dx + dy * dy + dz * dz);
    float zfar = distance + radius;
    float znear = fmaxf(distance - radius, zfar / 1000.0f);
    cam->zfar = zfar;
    cam->znear = znear;
}

/* src/renderer.rs:136-141 + 321-343:
 *   focal = projection.focal(viewport); viewport = viewport as f32;
 *   proj = VIEWPORT_Y_FLIP * proj_matrix; proj_inv = proj_matrix.invert();
 *   view = world2view(Matrix3::from(rotation), position); view_inv = view.invert() */
void wso_camera_uniform_build(const wso_camera* cam, uint32_t vw, uint32_t vh, wso_camera_uniform* u) {
    float r[9];
    wso_quat_to_mat3(cam->rotation, r);
    wso_world2view(r, cam->position, u->view);
    if (!mat4_invert(u->view, u->view_inv)) memset(u->view_inv, 0, sizeof u->view_inv);
    float proj[16];
    wso_build_proj(cam->znear, cam->zfar, cam->fovx, cam->fovy, proj);
    float flip[16];
    memset(flip, 0, sizeof flip);
    M4(flip, 0, 0) = 1.0f;
    M4(flip, 1, 1) = -1.0f;
    M4(flip, 2, 2) = 1.0f;
    M4(flip, 3, 3) = 1.0f;
    mat4_mul(flip, proj, u->proj);
    if (!mat4_invert(proj, u->proj_inv)) memset(u->proj_inv, 0, sizeof u->proj_inv);
    u->viewport[0] = (float)vw;
    u->viewport[1] = (float)vh;
    u->focal[0] = wso_fov2focal(cam->fovx, (float)vw);
    u->focal[1] = wso_fov2focal(cam->fovy, (float)vh);
}

/* src/scene.rs:85-108: impl Into<PerspectiveCamera> for SceneCamera */
void wso_scene_camera_to_perspective(const float position[3], const float rotation_rows[9], float fx, float fy,
                                     uint32_t width, uint32_t height, wso_camera* out) {
    float fovx = wso_focal2fov(fx, (float)width);
    float fovy = wso_focal2fov(fy, (float)height);
    /* Matrix3::from([[f32;3];3]) takes each inner array as a COLUMN */
    float rot[9];
    memcpy(rot, rotation_rows, sizeof rot);
    float det = det3(rot[0], rot[3], rot[6], rot[1], rot[4], rot[7], rot[2], rot[5], rot[8]);
    if (det < 0.0f) {
        rot[0 * 3 + 1] = -rot[0 * 3 + 1];
        rot[1 * 3 + 1] = -rot[1 * 3 + 1];
        rot[2 * 3 + 1] = -rot[2 * 3 + 1];
    }
    memcpy(out->position, position, 3 * sizeof(float));
    wso_mat3_to_quat(rot, out->rotation);
    out->fovx = fovx;
    out->fovy = fovy;
    out->znear = 0.01f;
    out->zfar = 100.0f;
    float vr = (float)width / (float)height;
    float fr = fovx / fovy;
    out->fov2view_ratio = vr / fr;
}

/* ------------------------------------------------------------------------- */
/* loader data prep                                                           */
/* ------------------------------------------------------------------------- */

/* src/utils.rs:206-212 */
float wso_sigmoid(float x) {
    if (x >= 0.0f) return 1.0f / (1.0f + expf(-x));
    return expf(x) / (1.0f + expf(x));
}

/* src/utils.rs:194-203: r = Matrix3::from(rot); l = r * diag(scale); m = l * l^T;
 * returns [m00, m01, m02, m11, m12, m22] (m[c][r], symmetric) */
void wso_build_cov(const float q[4], const float scale[3], float out[6]) {
    float r[9], l[9], m[9];
    wso_quat_to_mat3(q, r);
    /* r * s with s = from_diagonal(scale): cgmath Matrix3 * Matrix3 = row(rr).dot(column c of s), and
     * Vector3::dot = x*x' + y*y' + z*z' (no leading zero: the *0 terms decide the SIGN of an exact zero) */
    for (int c = 0; c < 3; c++)
        for (int rr = 0; rr < 3; rr++) {
            float s = M3(r, 0, rr) * (0 == c ? scale[c] : 0.0f);
            s += M3(r, 1, rr) * (1 == c ? scale[c] : 0.0f);
            s += M3(r, 2, rr) * (2 == c ? scale[c] : 0.0f);
            M3(l, c, rr) = s;
        }
    /* m = l * l^T : m[c][r] = sum_k l[k][r] * lT[c][k] = sum_k l[k][r] * l[k][c] */
    for (int c = 0; c < 3; c++)
        for (int rr = 0; rr < 3; rr++) {
            float s = M3(l, 0, rr) * M3(l, 0, c);
            s += M3(l, 1, rr) * M3(l, 1, c);
            s += M3(l, 2, rr) * M3(l, 2, c);
            M3(m, c, rr) = s;
        }
    out[0] = M3(m, 0, 0);
    out[1] = M3(m, 0, 1);
    out[2] = M3(m, 0, 2);
    out[3] = M3(m, 1, 1);
    out[4] = M3(m, 1, 2);
    out[5] = M3(m, 2, 2);
}

static void put_f32(uint8_t* p, float f) { memcpy(p, &f, 4); }
static void put_u16(uint8_t* p, uint16_t h) { memcpy(p, &h, 2); }
static float get_f32(const uint8_t* p) {
    float f;
    memcpy(&f, p, 4);
    return f;
}
static uint16_t get_u16(const uint8_t* p) {
    uint16_t h;
    memcpy(&h, p, 2);
    return h;
}
static uint32_t get_u32(const uint8_t* p) {
    uint32_t h;
    memcpy(&h, p, 4);
    return h;
}

/* src/io/ply.rs:50-100 read_line, for every row */
void wso_ply_rows_convert(const float* rows, uint32_t n, uint32_t sh_deg, uint8_t* gaussians, uint8_t* sh_out) {
    uint32_t num_coefs = (sh_deg + 1) * (sh_deg + 1);
    uint32_t row_len = 3 + 3 + 3 + (num_coefs - 1) * 3 + 1 + 3 + 4;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        const float* row = rows + (size_t)i * row_len;
        const float* pos = row;
        const float* dc = row + 6;
        const float* rest = row + 9;
        const float* tail = rest + (num_coefs - 1) * 3;
        float sh[16][3];
        memset(sh, 0, sizeof sh);
        sh[0][0] = dc[0];
        sh[0][1] = dc[1];
        sh[0][2] = dc[2];
        for (uint32_t c = 0; c + 1 < num_coefs; c++)
            for (uint32_t j = 0; j < 3; j++) sh[c + 1][j] = rest[j * (num_coefs - 1) + c];
        float opacity = wso_sigmoid(tail[0]);
        float scale[3] = {expf(tail[1]), expf(tail[2]), expf(tail[3])};
        float q[4] = {tail[4], tail[5], tail[6], tail[7]};
        /* cgmath InnerSpace::normalize = self * (1 / magnitude) */
        float mag = sqrtf(q[0] * q[0] + (q[1] * q[1] + q[2] * q[2] + q[3] * q[3]));
        float inv = 1.0f / mag;
        for (int k = 0; k < 4; k++) q[k] = q[k] * inv;
        float cov[6];
        wso_build_cov(q, scale, cov);
        uint8_t* g = gaussians + (size_t)i * 28;
        put_f32(g + 0, pos[0]);
        put_f32(g + 4, pos[1]);
        put_f32(g + 8, pos[2]);
        put_u16(g + 12, wso_f32_to_f16(opacity));
        put_u16(g + 14, 0);
        for (int k = 0; k < 6; k++) put_u16(g + 16 + 2 * k, wso_f32_to_f16(cov[k]));
        uint8_t* s = sh_out + (size_t)i * 96;
        for (int c = 0; c < 16; c++)
            for (int j = 0; j < 3; j++) put_u16(s + (c * 3 + j) * 2, wso_f32_to_f16(sh[c][j]));
    }
}

/* src/io/mod.rs:63-105 (bbox grow from `start`), 185-284 plane_from_points.
 * Sequential f32 accumulation like the reference. */
int wso_pointcloud_stats(const uint8_t* gaussians, uint32_t n, uint32_t stride, const wso_aabb* start,
                         wso_aabb* bbox, float center[3], float up[3]) {
    *bbox = *start;
    float sum[3] = {0, 0, 0};
    for (uint32_t i = 0; i < n; i++) {
        const uint8_t* g = gaussians + (size_t)i * stride;
        for (int k = 0; k < 3; k++) {
            float v = get_f32(g + 4 * k);
            bbox->min[k] = fminf(bbox->min[k], v);
            bbox->max[k] = fmaxf(bbox->max[k], v);
            sum[k] = sum[k] + v;
        }
    }
    float inv_n = 1.0f / (float)n;
    for (int k = 0; k < 3; k++) center[k] = sum[k] * inv_n;
    up[0] = up[1] = up[2] = 0.0f;
    if (n < 3) return 0;
    float xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
    for (uint32_t i = 0; i < n; i++) {
        const uint8_t* g = gaussians + (size_t)i * stride;
        float rx = get_f32(g) - center[0], ry = get_f32(g + 4) - center[1], rz = get_f32(g + 8) - center[2];
        xx += rx * rx;
        xy += rx * ry;
        xz += rx * rz;
        yy += ry * ry;
        yz += ry * rz;
        zz += rz * rz;
    }
    float fn = (float)n;
    xx /= fn;
    xy /= fn;
    xz /= fn;
    yy /= fn;
    yz /= fn;
    zz /= fn;
    float wd[3] = {0, 0, 0};
    {
        float det_x = yy * zz - yz * yz;
        float ax[3] = {det_x, xz * yz - xy * zz, xy * yz - xz * yy};
        float weight = det_x * det_x;
        if (wd[0] * ax[0] + wd[1] * ax[1] + wd[2] * ax[2] < 0.0f) weight = -weight;
        for (int k = 0; k < 3; k++) wd[k] += ax[k] * weight;
    }
    {
        float det_y = xx * zz - xz * xz;
        float ax[3] = {xz * yz - xy * zz, det_y, xy * xz - yz * xx};
        float weight = det_y * det_y;
        if (wd[0] * ax[0] + wd[1] * ax[1] + wd[2] * ax[2] < 0.0f) weight = -weight;
        for (int k = 0; k < 3; k++) wd[k] += ax[k] * weight;
    }
    {
        float det_z = xx * yy - xy * xy;
        float ax[3] = {xy * yz - xz * yy, xy * xz - yz * xx, det_z};
        float weight = det_z * det_z;
        if (wd[0] * ax[0] + wd[1] * ax[1] + wd[2] * ax[2] < 0.0f) weight = -weight;
        for (int k = 0; k < 3; k++) wd[k] += ax[k] * weight;
    }
    float mag = sqrtf(wd[0] * wd[0] + wd[1] * wd[1] + wd[2] * wd[2]);
    float inv = 1.0f / mag;
    float nrm[3] = {wd[0] * inv, wd[1] * inv, wd[2] * inv};
    if (nrm[1] < 0.0f) {
        nrm[0] = -nrm[0];
        nrm[1] = -nrm[1];
        nrm[2] = -nrm[2];
    }
    int finite = isfinite(nrm[0]) && isfinite(nrm[1]) && isfinite(nrm[2]);
    /* io/mod.rs:88-90: up is dropped for small scenes */
    if (wso_aabb_radius(bbox) < 10.0f) finite = 0;
    if (finite) {
        up[0] = nrm[0];
        up[1] = nrm[1];
        up[2] = nrm[2];
    }
    return finite;
}

/* ------------------------------------------------------------------------- */
/* K1 / K1c                                                                   */
/* ------------------------------------------------------------------------- */

static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                               -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

typedef struct {
    float c[16][3];
} sh_set;

/* preprocess.wgsl:124-154 evaluate_sh (vector expression evaluated per channel) */
static void evaluate_sh(const float dir[3], const sh_set* sh, uint32_t sh_deg, float out[3]) {
    float x = dir[0], y = dir[1], z = dir[2];
    for (int ch = 0; ch < 3; ch++) {
#define C(i) (sh->c[i][ch])
        float result = SH_C0 * C(0);
        if (sh_deg > 0u) {
            result += -SH_C1 * y * C(1) + SH_C1 * z * C(2) - SH_C1 * x * C(3);
            if (sh_deg > 1u) {
                float xx = x * x, yy = y * y, zz = z * z;
                float xy = x * y, yz = y * z, xz = x * z;
                result += SH_C2[0] * xy * C(4) + SH_C2[1] * yz * C(5) + SH_C2[2] * (2.0f * zz - xx - yy) * C(6) +
                          SH_C2[3] * xz * C(7) + SH_C2[4] * (xx - yy) * C(8);
                if (sh_deg > 2u) {
                    result += SH_C3[0] * y * (3.0f * xx - yy) * C(9) + SH_C3[1] * xy * z * C(10) +
                              SH_C3[2] * y * (4.0f * zz - xx - yy) * C(11) +
                              SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * C(12) +
                              SH_C3[4] * x * (4.0f * zz - xx - yy) * C(13) + SH_C3[5] * z * (xx - yy) * C(14) +
                              SH_C3[6] * x * (xx - 3.0f * yy) * C(15);
                }
            }
        }
#undef C
        result += 0.5f;
        out[ch] = result;
    }
}

static float smoothstep01(float x) {
    /* WGSL smoothstep(0,1,x): t = clamp((x - 0)/(1 - 0), 0, 1); t*t*(3 - 2t) */
    float t = (x - 0.0f) / (1.0f - 0.0f);
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    return t * t * (3.0f - 2.0f * t);
}

/* 3x3 column-major product (WGSL mat3x3 * mat3x3) */
static void mat3_mul(const float* a, const float* b, float* out) {
    float t[9];
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) {
            float s = M3(a, 0, r) * M3(b, c, 0);
            s += M3(a, 1, r) * M3(b, c, 1);
            s += M3(a, 2, r) * M3(b, c, 2);
            t[c * 3 + r] = s;
        }
    memcpy(out, t, sizeof t);
}
static void mat3_transpose(const float* a, float* out) {
    float t[9];
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) t[c * 3 + r] = M3(a, r, c);
    memcpy(out, t, sizeof t);
}

typedef struct {
    int visible;
    uint8_t splat[20];
    uint32_t key;
} k1_result;

/* shared tail of both preprocess variants, from the frustum test on.
 * compressed = 0: preprocess.wgsl:190-273; compressed = 1: preprocess_compressed.wgsl:231-325 */
static void k1_body(const float xyz[3], float opacity, const float cov6[6], const sh_set* sh,
                    const wso_camera_uniform* cam, const wso_settings_uniform* rs, int compressed, k1_result* res) {
    res->visible = 0;
    const float* view = cam->view;
    const float* proj = cam->proj;
    /* camspace = view * vec4(xyz, 1) */
    float camspace[4], pos2d[4];
    for (int r = 0; r < 4; r++) {
        float s = M4(view, 0, r) * xyz[0];
        s += M4(view, 1, r) * xyz[1];
        s += M4(view, 2, r) * xyz[2];
        s += M4(view, 3, r) * 1.0f;
        camspace[r] = s;
    }
    for (int r = 0; r < 4; r++) {
        float s = M4(proj, 0, r) * camspace[0];
        s += M4(proj, 1, r) * camspace[1];
        s += M4(proj, 2, r) * camspace[2];
        s += M4(proj, 3, r) * camspace[3];
        pos2d[r] = s;
    }
    float bounds = 1.2f * pos2d[3];
    float z = pos2d[2] / pos2d[3];
    if (!compressed) {
        if (z <= 0.0f || z >= 1.0f || pos2d[0] < -bounds || pos2d[0] > bounds || pos2d[1] < -bounds ||
            pos2d[1] > bounds)
            return;
    } else {
        if (z < 0.0f || z > 1.0f || pos2d[0] < -bounds || pos2d[0] > bounds || pos2d[1] < -bounds ||
            pos2d[1] > bounds)
            return;
    }
    /* NaN z (w == 0 and z == 0) passes neither test in WGSL nor here: comparisons are false -> kept,
     * as in the reference. */

    float walltime = rs->walltime;
    float scale_mod = 0.0f;
    float ddx = rs->scene_center[0] - xyz[0], ddy = rs->scene_center[1] - xyz[1], ddz = rs->scene_center[2] - xyz[2];
    float dd = 5.0f * sqrtf(ddx * ddx + ddy * ddy + ddz * ddz) / rs->scene_extend;
    if (walltime > dd) scale_mod = smoothstep01(walltime - dd);
    float scaling = rs->gaussian_scaling * scale_mod;

    float Vrk[9] = {cov6[0], cov6[1], cov6[2], cov6[1], cov6[3], cov6[4], cov6[2], cov6[4], cov6[5]};
    for (int i = 0; i < 9; i++) Vrk[i] = Vrk[i] * scaling * scaling;
    float fx = cam->focal[0], fy = cam->focal[1];
    float J[9] = {fx / camspace[2],
                  0.0f,
                  -(fx * camspace[0]) / (camspace[2] * camspace[2]),
                  0.0f,
                  -fy / camspace[2],
                  (fy * camspace[1]) / (camspace[2] * camspace[2]),
                  0.0f,
                  0.0f,
                  0.0f};
    float Wm[9] = {M4(view, 0, 0), M4(view, 0, 1), M4(view, 0, 2), M4(view, 1, 0), M4(view, 1, 1),
                   M4(view, 1, 2), M4(view, 2, 0), M4(view, 2, 1), M4(view, 2, 2)};
    float W[9], T[9], Tt[9], tmp[9], cov[9];
    mat3_transpose(Wm, W);
    mat3_mul(W, J, T);
    mat3_transpose(T, Tt);
    mat3_mul(Tt, Vrk, tmp);
    mat3_mul(tmp, T, cov);

    float kernel_size = rs->kernel_size;
    if (rs->mip_splatting != 0u) {
        float det_0 = fmaxf(1e-6f, M3(cov, 0, 0) * M3(cov, 1, 1) - M3(cov, 0, 1) * M3(cov, 0, 1));
        float det_1 = fmaxf(1e-6f, (M3(cov, 0, 0) + kernel_size) * (M3(cov, 1, 1) + kernel_size) -
                                       M3(cov, 0, 1) * M3(cov, 0, 1));
        float coef = sqrtf(det_0 / (det_1 + 1e-6f) + 1e-6f);
        if (det_0 <= 1e-6f || det_1 <= 1e-6f) coef = 0.0f;
        opacity *= coef;
    }
    float diagonal1 = M3(cov, 0, 0) + kernel_size;
    float offDiagonal = M3(cov, 0, 1);
    float diagonal2 = M3(cov, 1, 1) + kernel_size;
    float mid = 0.5f * (diagonal1 + diagonal2);
    float hx = (diagonal1 - diagonal2) / 2.0f;
    float radius = sqrtf(hx * hx + offDiagonal * offDiagonal);
    float lambda1, lambda2;
    if (!compressed) {
        lambda1 = mid + radius;
        lambda2 = fmaxf(mid - radius, 0.1f);
    } else {
        lambda1 = mid + fmaxf(radius, 0.1f);
        lambda2 = mid - fmaxf(radius, 0.1f);
    }
    float dvx = offDiagonal, dvy = lambda1 - diagonal1;
    float dlen = sqrtf(dvx * dvx + dvy * dvy);
    float ex, ey;
    if (dlen > 0.0f) {
        ex = dvx / dlen;
        ey = dvy / dlen;
    } else {
        /* WGSL normalize((0,0)) is undefined (NaN on most drivers).  Project decision:
         * define the axis as (1,0); tests exclude such splats from bit parity claims. */
        ex = 1.0f;
        ey = 0.0f;
    }
    float s1 = sqrtf(2.0f * lambda1), s2 = sqrtf(2.0f * lambda2);
    float v1x = s1 * ex, v1y = s1 * ey;
    float v2x = s2 * ey, v2y = s2 * (-ex);
    float vcx = pos2d[0] / pos2d[3], vcy = pos2d[1] / pos2d[3];

    const float* vinv = cam->view_inv;
    float dx = xyz[0] - M4(vinv, 3, 0), dy = xyz[1] - M4(vinv, 3, 1), dz = xyz[2] - M4(vinv, 3, 2);
    float dl = sqrtf(dx * dx + dy * dy + dz * dz);
    float dir[3] = {dx / dl, dy / dl, dz / dl};
    float col[3];
    evaluate_sh(dir, sh, rs->max_sh_deg, col);
    for (int k = 0; k < 3; k++) col[k] = fmaxf(0.0f, col[k]);

    float vw = cam->viewport[0], vh = cam->viewport[1];
    uint16_t h[10] = {wso_f32_to_f16(v1x / vw), wso_f32_to_f16(v1y / vh), wso_f32_to_f16(v2x / vw),
                      wso_f32_to_f16(v2y / vh), wso_f32_to_f16(vcx),      wso_f32_to_f16(vcy),
                      wso_f32_to_f16(col[0]),   wso_f32_to_f16(col[1]),   wso_f32_to_f16(col[2]),
                      wso_f32_to_f16(opacity)};
    memcpy(res->splat, h, 20);

    float znear = -M4(proj, 3, 2) / M4(proj, 2, 2);
    float zfar = -M4(proj, 3, 2) / (M4(proj, 2, 2) - 1.0f);
    if (!compressed) {
        res->key = f32_bits(zfar - pos2d[2]);
    } else {
        float kf = 16777215.0f - (pos2d[2] - znear) / (zfar - znear) * 16777215.0f;
        /* WGSL u32(f32) saturates */
        if (!(kf > 0.0f))
            res->key = 0u;
        else if (kf >= 4294967296.0f)
            res->key = 0xFFFFFFFFu;
        else
            res->key = (uint32_t)kf;
    }
    res->visible = 1;
}

static int clip_reject(const float xyz[3], const wso_settings_uniform* rs) {
    /* preprocess.wgsl:177-179 */
    for (int k = 0; k < 3; k++)
        if (xyz[k] < rs->clip_min[k] || xyz[k] > rs->clip_max[k]) return 1;
    return 0;
}

static uint32_t compact(const k1_result* res, uint32_t n, uint8_t* splats, uint32_t* keys, uint32_t* src_index) {
    uint32_t v = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (!res[i].visible) continue;
        if (splats) memcpy(splats + (size_t)v * 20, res[i].splat, 20);
        if (keys) keys[v] = res[i].key;
        if (src_index) src_index[v] = i;
        v++;
    }
    return v;
}

uint32_t wso_preprocess(const uint8_t* gaussians, const uint8_t* sh, uint32_t n, const wso_camera_uniform* cam,
                        const wso_settings_uniform* rs, uint8_t* splats, uint32_t* keys, uint32_t* src_index) {
    k1_result* res = (k1_result*)malloc(sizeof(k1_result) * (n ? n : 1));
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        const uint8_t* g = gaussians + (size_t)i * 28;
        float xyz[3] = {get_f32(g), get_f32(g + 4), get_f32(g + 8)};
        res[i].visible = 0;
        if (clip_reject(xyz, rs)) continue;
        float opacity = wso_f16_to_f32(get_u16(g + 12));
        float cov6[6];
        for (int k = 0; k < 6; k++) cov6[k] = wso_f16_to_f32(get_u16(g + 16 + 2 * k));
        sh_set s;
        const uint8_t* sp = sh + (size_t)i * 96;
        for (int c = 0; c < 16; c++)
            for (int j = 0; j < 3; j++) s.c[c][j] = wso_f16_to_f32(get_u16(sp + (c * 3 + j) * 2));
        k1_body(xyz, opacity, cov6, &s, cam, rs, 0, &res[i]);
    }
    uint32_t v = compact(res, n, splats, keys, src_index);
    free(res);
    return v;
}

/* preprocess_compressed.wgsl:137-171: dequantize + sh_coef.
 * unpack4x8snorm(b) = max(i8/127, -1); times 127 -> i8 clamped at -127. */
static float snorm_times_127(int8_t b) {
    float v = fmaxf((float)b / 127.0f, -1.0f);
    return v * 127.0f;
}

uint32_t wso_preprocess_compressed(const uint8_t* gaussians, const uint8_t* sh_bytes, const uint8_t* covars,
                                   const wso_gaussian_quantization* q, uint32_t n, uint32_t sh_deg_layout,
                                   const wso_camera_uniform* cam, const wso_settings_uniform* rs, uint8_t* splats,
                                   uint32_t* keys, uint32_t* src_index) {
    k1_result* res = (k1_result*)malloc(sizeof(k1_result) * (n ? n : 1));
    uint32_t ncoef = (sh_deg_layout + 1) * (sh_deg_layout + 1);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        const uint8_t* g = gaussians + (size_t)i * 24;
        float xyz[3] = {get_f32(g), get_f32(g + 4), get_f32(g + 8)};
        res[i].visible = 0;
        if (clip_reject(xyz, rs)) continue;
        int8_t op_i8 = (int8_t)g[12];
        int8_t sc_i8 = (int8_t)g[13];
        uint32_t geometry_idx = get_u32(g + 16);
        uint32_t sh_idx = get_u32(g + 20);
        /* dequantize(value, q) = (f32(value) - f32(q.zero_point)) * q.scaling  (line 137-139) */
        float opacity = ((float)op_i8 - (float)q->opacity.zero_point) * q->opacity.scale;
        float scaling_factor = expf(((float)sc_i8 - (float)q->scaling_factor.zero_point) * q->scaling_factor.scale);
        float s2 = scaling_factor * scaling_factor;
        const uint8_t* cv = covars + (size_t)geometry_idx * 12;
        float cov6[6];
        for (int k = 0; k < 6; k++) cov6[k] = wso_f16_to_f32(get_u16(cv + 2 * k)) * s2;
        sh_set s;
        memset(&s, 0, sizeof s);
        uint32_t use_coefs = (rs->max_sh_deg + 1) * (rs->max_sh_deg + 1);
        if (use_coefs > ncoef) use_coefs = ncoef; /* reads past the record are a caller error in the reference */
        for (uint32_t c = 0; c < use_coefs; c++) {
            const wso_quantization* qq = (c == 0) ? &q->color_dc : &q->color_rest;
            size_t off = (size_t)3 * ((size_t)sh_idx * ncoef + c);
            for (int j = 0; j < 3; j++) {
                float v = snorm_times_127((int8_t)sh_bytes[off + j]);
                s.c[c][j] = (v - (float)qq->zero_point) * qq->scale;
            }
        }
        k1_body(xyz, opacity, cov6, &s, cam, rs, 1, &res[i]);
    }
    uint32_t v = compact(res, n, splats, keys, src_index);
    free(res);
    return v;
}

/* ------------------------------------------------------------------------- */
/* sort contract                                                              */
/* ------------------------------------------------------------------------- */

/* gpu_rs.rs:865-884 + radix_sort.wgsl: 4 passes x 8-bit digits, LSD, stable, ascending. */
void wso_sort_pairs(uint32_t* keys, uint32_t* payload, uint32_t n) {
    if (n == 0) return;
    uint32_t* k2 = (uint32_t*)malloc(sizeof(uint32_t) * n);
    uint32_t* p2 = (uint32_t*)malloc(sizeof(uint32_t) * n);
    uint32_t *ka = keys, *pa = payload, *kb = k2, *pb = p2;
    for (int pass = 0; pass < 4; pass++) {
        size_t hist[256];
        memset(hist, 0, sizeof hist);
        int shift = pass * 8;
        for (uint32_t i = 0; i < n; i++) hist[(ka[i] >> shift) & 0xFFu]++;
        size_t sum = 0;
        for (int d = 0; d < 256; d++) {
            size_t c = hist[d];
            hist[d] = sum;
            sum += c;
        }
        for (uint32_t i = 0; i < n; i++) {
            size_t dst = hist[(ka[i] >> shift) & 0xFFu]++;
            kb[dst] = ka[i];
            pb[dst] = pa[i];
        }
        uint32_t* t = ka;
        ka = kb;
        kb = t;
        t = pa;
        pa = pb;
        pb = t;
    }
    /* 4 passes: result is back in the caller's buffers (A -> B -> A -> B -> A) */
    free(k2);
    free(p2);
}

/* ------------------------------------------------------------------------- */
/* K6: quad rasterisation + premultiplied-alpha blending                      */
/* ------------------------------------------------------------------------- */

static const float CUTOFF = 2.3539888583335364f;

static float quantize_target(float v, int mode) {
    if (mode == 1) return wso_f16_to_f32(wso_f32_to_f16(v));
    if (mode == 2) {
        /* unorm8 render target: clamp, scale, round to nearest (ties to even per D3D/Vulkan rules) */
        float c = fminf(fmaxf(v, 0.0f), 1.0f);
        return nearbyintf(c * 255.0f) / 255.0f;
    }
    return v;
}

typedef struct {
    float cx, cy;               /* centre, pixels (x right, y down) */
    float i00, i01, i10, i11;   /* screen_pos = I * (pixel - centre) */
    float r, g, b, a;
    int x0, x1, y0, y1;         /* conservative pixel bbox, inclusive */
    int valid;
} raster_splat;

/* gaussian.wgsl:30-56 vs_main: quad corners position = (+-1,+-1)*CUTOFF,
 * ndc = v_center + 2 * mat2x2(v1, v2) * position;  screen_pos = position is interpolated affinely.
 * WebGPU viewport transform: px = (ndc.x*0.5+0.5)*W, py = (0.5-ndc.y*0.5)*H. */
static void setup_splat(const uint8_t* sp, uint32_t w, uint32_t h, raster_splat* rs) {
    float v1x = wso_f16_to_f32(get_u16(sp + 0)), v1y = wso_f16_to_f32(get_u16(sp + 2));
    float v2x = wso_f16_to_f32(get_u16(sp + 4)), v2y = wso_f16_to_f32(get_u16(sp + 6));
    float pcx = wso_f16_to_f32(get_u16(sp + 8)), pcy = wso_f16_to_f32(get_u16(sp + 10));
    rs->r = wso_f16_to_f32(get_u16(sp + 12));
    rs->g = wso_f16_to_f32(get_u16(sp + 14));
    rs->b = wso_f16_to_f32(get_u16(sp + 16));
    rs->a = wso_f16_to_f32(get_u16(sp + 18));
    float W = (float)w, H = (float)h;
    /* pixel offset d = M * position, M columns = images of the unit position axes */
    float m00 = v1x * W, m01 = v2x * W;   /* d.x = m00*p.x + m01*p.y */
    float m10 = -v1y * H, m11 = -v2y * H; /* d.y (down) */
    float det = m00 * m11 - m01 * m10;
    rs->valid = 0;
    if (!(fabsf(det) > 0.0f) || !isfinite(det)) return;
    float inv = 1.0f / det;
    rs->i00 = m11 * inv;
    rs->i01 = -m01 * inv;
    rs->i10 = -m10 * inv;
    rs->i11 = m00 * inv;
    rs->cx = (pcx * 0.5f + 0.5f) * W;
    rs->cy = (0.5f - pcy * 0.5f) * H;
    if (!isfinite(rs->cx) || !isfinite(rs->cy)) return;
    /* |position|^2 <= 2*CUTOFF -> |d.x| <= sqrt(2*CUTOFF)*|row_x(M)| ; pad generously, exact test per pixel */
    float rad = sqrtf(2.0f * CUTOFF);
    float ex = rad * sqrtf(m00 * m00 + m01 * m01) * 1.01f + 1.0f;
    float ey = rad * sqrtf(m10 * m10 + m11 * m11) * 1.01f + 1.0f;
    if (!isfinite(ex) || !isfinite(ey)) return;
    float fx0 = floorf(rs->cx - ex), fx1 = ceilf(rs->cx + ex);
    float fy0 = floorf(rs->cy - ey), fy1 = ceilf(rs->cy + ey);
    if (fx1 < 0.0f || fy1 < 0.0f || fx0 > W - 1.0f || fy0 > H - 1.0f) return;
    rs->x0 = (int)fmaxf(fx0, 0.0f);
    rs->y0 = (int)fmaxf(fy0, 0.0f);
    rs->x1 = (int)fminf(fx1, W - 1.0f);
    rs->y1 = (int)fminf(fy1, H - 1.0f);
    rs->valid = 1;
}

void wso_render(const uint8_t* splats, const uint32_t* sorted_indices, uint32_t v, uint32_t w, uint32_t h,
                const float background[4], int target_mode, float* out) {
    raster_splat* rsp = (raster_splat*)malloc(sizeof(raster_splat) * (v ? v : 1));
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)v; i++) {
        uint32_t idx = sorted_indices ? sorted_indices[i] : (uint32_t)i;
        setup_splat(splats + (size_t)idx * 20, w, h, &rsp[i]);
    }
    const int band = 16;
    int nbands = ((int)h + band - 1) / band;
#pragma omp parallel for schedule(dynamic, 1)
    for (int bi = 0; bi < nbands; bi++) {
        int by0 = bi * band, by1 = by0 + band - 1;
        if (by1 > (int)h - 1) by1 = (int)h - 1;
        for (int y = by0; y <= by1; y++)
            for (uint32_t x = 0; x < w; x++)
                for (int c = 0; c < 4; c++)
                    out[((size_t)y * w + x) * 4 + c] = quantize_target(background[c], target_mode);
        /* instances in sorted order = far -> near (ascending key); blend: dst = src + dst*(1 - src.a) */
        for (uint32_t i = 0; i < v; i++) {
            const raster_splat* s = &rsp[i];
            if (!s->valid || s->y1 < by0 || s->y0 > by1) continue;
            int ya = s->y0 > by0 ? s->y0 : by0, yb = s->y1 < by1 ? s->y1 : by1;
            for (int y = ya; y <= yb; y++) {
                float dy = ((float)y + 0.5f) - s->cy;
                for (int x = s->x0; x <= s->x1; x++) {
                    float dx = ((float)x + 0.5f) - s->cx;
                    float p0 = s->i00 * dx + s->i01 * dy;
                    float p1 = s->i10 * dx + s->i11 * dy;
                    float a = p0 * p0 + p1 * p1; /* gaussian.wgsl:60 dot(screen_pos, screen_pos) */
                    if (!(a <= 2.0f * CUTOFF)) continue; /* :61 discard if a > 2*CUTOFF */
                    float b = fminf(0.99f, expf(-a) * s->a);
                    float* px = out + ((size_t)y * w + x) * 4;
                    float one_minus = 1.0f - b;
                    px[0] = quantize_target(s->r * b + px[0] * one_minus, target_mode);
                    px[1] = quantize_target(s->g * b + px[1] * one_minus, target_mode);
                    px[2] = quantize_target(s->b * b + px[2] * one_minus, target_mode);
                    px[3] = quantize_target(1.0f * b + px[3] * one_minus, target_mode);
                }
            }
        }
    }
    free(rsp);
}
