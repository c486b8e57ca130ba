// host_math.cpp -- host side of the drop-in boundary: camera matrices, uniforms, loader data prep.
//
// Mirrors (reference file:line under /root/reference):
//   src/camera.rs:26-35 fit_near_far, 146-152 focal, 207-214 world2view, 216-234 build_proj,
//   236-242 focal2fov / fov2focal; src/renderer.rs:321-343 CameraUniform::set_*,
//   620-651 SplattingArgsUniform::from_args_and_pc; src/scene.rs:85-108 SceneCamera -> PerspectiveCamera;
//   src/io/ply.rs:50-100 read_line; src/utils.rs:194-212 build_cov / sigmoid;
//   src/io/mod.rs:63-105 bbox + centroid, 185-284 plane_from_points.
// cgmath (git ff840cbf) and half 2.6.0 are not vendored in the reference; their published
// algorithms (quaternion->matrix, cofactor inverse, Shoemake matrix->quaternion, RTE f16) are
// written out here.  Compiled with -ffp-contract=off so the f32 operation order is the source order.
#include <cmath>
#include <cstring>

#include "ws_internal.h"

namespace ws {

namespace {

struct Mat4 {
    float m[16];  // column-major: m[c*4 + r]
    float& at(int c, int r) { return m[c * 4 + r]; }
    float at(int c, int r) const { return m[c * 4 + r]; }
};
struct Mat3 {
    float m[9];  // column-major
    float& at(int c, int r) { return m[c * 3 + r]; }
    float at(int c, int r) const { return m[c * 3 + r]; }
};

float det3x3(const float a[3][3]) {
    return a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
           a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
}

// cofactor inverse (cgmath SquareMatrix::invert for Matrix4)
bool invert(const Mat4& a, Mat4* out) {
    float cof[4][4];  // cof[c][r]
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            float sub[3][3];
            int k = 0;
            for (int cc = 0; cc < 4; ++cc) {
                if (cc == c) continue;
                int j = 0;
                for (int rr = 0; rr < 4; ++rr) {
                    if (rr == r) continue;
                    sub[k][j++] = a.at(cc, rr);
                }
                ++k;
            }
            float d = det3x3(sub);
            cof[c][r] = ((r + c) & 1) ? -d : d;
        }
    float det = a.at(0, 0) * cof[0][0] + a.at(1, 0) * cof[1][0] + a.at(2, 0) * cof[2][0] + a.at(3, 0) * cof[3][0];
    if (det == 0.0f) return false;
    float inv_det = 1.0f / det;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) out->at(c, r) = cof[r][c] * inv_det;
    return true;
}

Mat4 transpose(const Mat4& a) {
    Mat4 t;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) t.at(c, r) = a.at(r, c);
    return t;
}

Mat4 mul(const Mat4& a, const Mat4& b) {
    Mat4 o;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            float s = a.at(0, r) * b.at(c, 0);
            s += a.at(1, r) * b.at(c, 1);
            s += a.at(2, r) * b.at(c, 2);
            s += a.at(3, r) * b.at(c, 3);
            o.at(c, r) = s;
        }
    return o;
}

// cgmath: impl From<Quaternion<S>> for Matrix3<S>; q = (s, x, y, z)
Mat3 quat_to_mat3(const float q[4]) {
    const float s = q[0], x = q[1], y = q[2], z = q[3];
    const float x2 = x + x, y2 = y + y, z2 = z + z;
    const float xx2 = x2 * x, xy2 = x2 * y, xz2 = x2 * z;
    const float yy2 = y2 * y, yz2 = y2 * z, zz2 = z2 * z;
    const float sy2 = y2 * s, sz2 = z2 * s, sx2 = x2 * s;
    Mat3 m;
    m.at(0, 0) = 1.0f - yy2 - zz2;
    m.at(0, 1) = xy2 + sz2;
    m.at(0, 2) = xz2 - sy2;
    m.at(1, 0) = xy2 - sz2;
    m.at(1, 1) = 1.0f - xx2 - zz2;
    m.at(1, 2) = yz2 + sx2;
    m.at(2, 0) = xz2 + sy2;
    m.at(2, 1) = yz2 - sx2;
    m.at(2, 2) = 1.0f - xx2 - yy2;
    return m;
}

// cgmath: impl From<Matrix3<S>> for Quaternion<S>
void mat3_to_quat(const Mat3& m, float q[4]) {
    const float trace = m.at(0, 0) + m.at(1, 1) + m.at(2, 2);
    float w, x, y, z;
    if (trace >= 0.0f) {
        float s = std::sqrt(1.0f + trace);
        w = 0.5f * s;
        s = 0.5f / s;
        x = (m.at(1, 2) - m.at(2, 1)) * s;
        y = (m.at(2, 0) - m.at(0, 2)) * s;
        z = (m.at(0, 1) - m.at(1, 0)) * s;
    } else if (m.at(0, 0) > m.at(1, 1) && m.at(0, 0) > m.at(2, 2)) {
        float s = std::sqrt((m.at(0, 0) - m.at(1, 1) - m.at(2, 2)) + 1.0f);
        x = 0.5f * s;
        s = 0.5f / s;
        y = (m.at(1, 0) + m.at(0, 1)) * s;
        z = (m.at(0, 2) + m.at(2, 0)) * s;
        w = (m.at(1, 2) - m.at(2, 1)) * s;
    } else if (m.at(1, 1) > m.at(2, 2)) {
        float s = std::sqrt((m.at(1, 1) - m.at(0, 0) - m.at(2, 2)) + 1.0f);
        y = 0.5f * s;
        s = 0.5f / s;
        z = (m.at(2, 1) + m.at(1, 2)) * s;
        x = (m.at(1, 0) + m.at(0, 1)) * s;
        w = (m.at(2, 0) - m.at(0, 2)) * s;
    } else {
        float s = std::sqrt((m.at(2, 2) - m.at(0, 0) - m.at(1, 1)) + 1.0f);
        z = 0.5f * s;
        s = 0.5f / s;
        x = (m.at(0, 2) + m.at(2, 0)) * s;
        y = (m.at(2, 1) + m.at(1, 2)) * s;
        w = (m.at(0, 1) - m.at(1, 0)) * s;
    }
    q[0] = w;
    q[1] = x;
    q[2] = y;
    q[3] = z;
}

// camera.rs:207-214
Mat4 world2view(const Mat3& r, const float t[3]) {
    Mat4 rt;
    std::memset(&rt, 0, sizeof rt);
    for (int c = 0; c < 3; ++c)
        for (int rr = 0; rr < 3; ++rr) rt.at(c, rr) = r.at(c, rr);
    rt.at(0, 3) = t[0];  // rt[0].w
    rt.at(1, 3) = t[1];
    rt.at(2, 3) = t[2];
    rt.at(3, 3) = 1.0f;
    Mat4 inv;
    if (!invert(rt, &inv)) std::memset(&inv, 0, sizeof inv);
    return transpose(inv);
}

// camera.rs:216-234
Mat4 build_proj(float znear, float zfar, float fov_x, float fov_y) {
    const float tan_half_fov_y = std::tan(fov_y / 2.0f);
    const float tan_half_fov_x = std::tan(fov_x / 2.0f);
    const float top = tan_half_fov_y * znear;
    const float bottom = -top;
    const float right = tan_half_fov_x * znear;
    const float left = -right;
    Mat4 p;
    std::memset(&p, 0, sizeof p);
    p.at(0, 0) = 2.0f * znear / (right - left);
    p.at(1, 1) = 2.0f * znear / (top - bottom);
    p.at(0, 2) = (right + left) / (right - left);
    p.at(1, 2) = (top + bottom) / (top - bottom);
    p.at(3, 2) = 1.0f;
    p.at(2, 2) = zfar / (zfar - znear);
    p.at(2, 3) = -(zfar * znear) / (zfar - znear);
    return transpose(p);
}

float fov2focal(float fov, float pixels) { return pixels / (2.0f * std::tan(fov * 0.5f)); }
float focal2fov(float focal, float pixels) { return 2.0f * std::atan(pixels / (2.0f * focal)); }

float sigmoid(float x) {  // utils.rs:206-212
    if (x >= 0.0f) return 1.0f / (1.0f + std::exp(-x));
    return std::exp(x) / (1.0f + std::exp(x));
}

void build_cov_impl(const float q[4], const float scale[3], float out[6]);

// utils.rs:194-203 build_cov: l = R * diag(s); m = l * l^T; upper triangle
void build_cov_impl(const float q[4], const float scale[3], float out[6]) {
    const Mat3 r = quat_to_mat3(q);
    Mat3 l;
    for (int c = 0; c < 3; ++c)
        for (int rr = 0; rr < 3; ++rr) {
            // cgmath Matrix3 * Matrix3 = row.dot(column), Vector3::dot = x*x' + y*y' + z*z': the full product with the
            // diagonal scale matrix, zero terms included and no leading zero (they decide the sign of an exact 0)
            float s = r.at(0, rr) * (0 == c ? scale[c] : 0.0f);
            s += r.at(1, rr) * (1 == c ? scale[c] : 0.0f);
            s += r.at(2, rr) * (2 == c ? scale[c] : 0.0f);
            l.at(c, rr) = s;
        }
    Mat3 m;
    for (int c = 0; c < 3; ++c)
        for (int rr = 0; rr < 3; ++rr) {
            float s = l.at(0, rr) * l.at(0, c);
            s += l.at(1, rr) * l.at(1, c);
            s += l.at(2, rr) * l.at(2, c);
            m.at(c, rr) = s;
        }
    out[0] = m.at(0, 0);
    out[1] = m.at(0, 1);
    out[2] = m.at(0, 2);
    out[3] = m.at(1, 1);
    out[4] = m.at(1, 2);
    out[5] = m.at(2, 2);
}

}  // namespace

void build_cov(const float q[4], const float scale[3], float out[6]) { build_cov_impl(q, scale, out); }

// IEEE binary16, round-to-nearest-even (half::f16::from_f32)
uint16_t host_f32_to_f16(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t exp = (x >> 23) & 0xFFu;
    uint32_t man = x & 0x7FFFFFu;
    if (exp == 0xFFu) return static_cast<uint16_t>(man ? (sign | 0x7E00u | (man >> 13)) : (sign | 0x7C00u));
    const int32_t e = static_cast<int32_t>(exp) - 112;
    if (e >= 31) return static_cast<uint16_t>(sign | 0x7C00u);
    if (e <= 0) {
        if (e < -10) return static_cast<uint16_t>(sign);
        man |= 0x800000u;
        const uint32_t shift = static_cast<uint32_t>(14 - e);
        uint32_t hm = man >> shift;
        const uint32_t rem = man & ((1u << shift) - 1u);
        const uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (hm & 1u))) ++hm;
        return static_cast<uint16_t>(sign | hm);
    }
    uint32_t out = sign | (static_cast<uint32_t>(e) << 10) | (man >> 13);
    const uint32_t rem = man & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (out & 1u))) ++out;
    return static_cast<uint16_t>(out);
}

float host_f16_to_f32(uint16_t h) {
    const uint32_t sign = (static_cast<uint32_t>(h) & 0x8000u) << 16;
    const uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {
            int e = -1;
            do {
                ++e;
                man <<= 1;
            } while (!(man & 0x400u));
            bits = sign | (static_cast<uint32_t>(112 - e) << 23) | ((man & 0x3FFu) << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | (man << 13);
    } else {
        bits = sign | ((exp + 112) << 23) | (man << 13);
    }
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

// renderer.rs:136-141 (focal, viewport) + 321-343 (set_camera)
void build_camera_uniform(const ws_camera& cam, const uint32_t viewport[2], ws_camera_uniform* out) {
    const Mat3 r = quat_to_mat3(cam.rotation);
    const Mat4 view = world2view(r, cam.position);
    Mat4 view_inv;
    if (!invert(view, &view_inv)) std::memset(&view_inv, 0, sizeof view_inv);
    const Mat4 proj = build_proj(cam.znear, cam.zfar, cam.fovx, cam.fovy);
    Mat4 flip;  // camera.rs:107-112 VIEWPORT_Y_FLIP
    std::memset(&flip, 0, sizeof flip);
    flip.at(0, 0) = 1.0f;
    flip.at(1, 1) = -1.0f;
    flip.at(2, 2) = 1.0f;
    flip.at(3, 3) = 1.0f;
    const Mat4 proj_flipped = mul(flip, proj);
    Mat4 proj_inv;  // inverse of the UN-flipped projection (renderer.rs:327-330)
    if (!invert(proj, &proj_inv)) std::memset(&proj_inv, 0, sizeof proj_inv);
    std::memcpy(out->view, view.m, sizeof view.m);
    std::memcpy(out->view_inv, view_inv.m, sizeof view_inv.m);
    std::memcpy(out->proj, proj_flipped.m, sizeof proj_flipped.m);
    std::memcpy(out->proj_inv, proj_inv.m, sizeof proj_inv.m);
    out->viewport[0] = static_cast<float>(viewport[0]);
    out->viewport[1] = static_cast<float>(viewport[1]);
    out->focal[0] = fov2focal(cam.fovx, static_cast<float>(viewport[0]));
    out->focal[1] = fov2focal(cam.fovy, static_cast<float>(viewport[1]));
}

}  // namespace ws

using namespace ws;

extern "C" {

float ws_aabb_radius(const ws_aabb* b) {
    if (!b) return 0.0f;
    const float dx = b->min[0] - b->max[0], dy = b->min[1] - b->max[1], dz = b->min[2] - b->max[2];
    return std::sqrt(dx * dx + dy * dy + dz * dz) / 2.0f;
}

int ws_camera_fit_near_far(ws_camera* cam, const ws_aabb* bbox) {
    if (!cam || !bbox) return fail(WS_ERR_INVALID, "ws_camera_fit_near_far: null argument");
    float c[3];
    for (int i = 0; i < 3; ++i) c[i] = bbox->min[i] + (bbox->max[i] - bbox->min[i]) / 2.0f;  // midpoint
    const float radius = ws_aabb_radius(bbox);
    const float dx = cam->position[0] - c[0], dy = cam->position[1] - c[1], dz = cam->position[2] - c[2];
    const float distance = std::sqrt(dx * dx + dy * dy + dz * dz);
    const float zfar = distance + radius;
    const float znear = std::fmax(distance - radius, zfar / 1000.0f);
    cam->zfar = zfar;
    cam->znear = znear;
    return WS_OK;
}

int ws_camera_from_scene(const float position[3], const float rotation[9], float fx, float fy, uint32_t width,
                         uint32_t height, ws_camera* out) {
    if (!position || !rotation || !out || width == 0 || height == 0)
        return fail(WS_ERR_INVALID, "ws_camera_from_scene: bad argument");
    const float fovx = focal2fov(fx, static_cast<float>(width));
    const float fovy = focal2fov(fy, static_cast<float>(height));
    Mat3 rot;  // Matrix3::from([[f32;3];3]): every inner array is a column
    std::memcpy(rot.m, rotation, sizeof rot.m);
    const float a[3][3] = {{rot.at(0, 0), rot.at(1, 0), rot.at(2, 0)},
                           {rot.at(0, 1), rot.at(1, 1), rot.at(2, 1)},
                           {rot.at(0, 2), rot.at(1, 2), rot.at(2, 2)}};
    if (det3x3(a) < 0.0f) {  // scene.rs:90-96: flip the y axis
        rot.at(0, 1) = -rot.at(0, 1);
        rot.at(1, 1) = -rot.at(1, 1);
        rot.at(2, 1) = -rot.at(2, 1);
    }
    std::memcpy(out->position, position, 3 * sizeof(float));
    mat3_to_quat(rot, out->rotation);
    out->fovx = fovx;
    out->fovy = fovy;
    out->znear = 0.01f;
    out->zfar = 100.0f;
    const float vr = static_cast<float>(width) / static_cast<float>(height);  // camera.rs:119-133
    const float fr = fovx / fovy;
    out->fov2view_ratio = vr / fr;
    return WS_OK;
}

int ws_build_camera_uniform(const ws_camera* cam, const uint32_t viewport[2], ws_camera_uniform* out) {
    if (!cam || !viewport || !out) return fail(WS_ERR_INVALID, "ws_build_camera_uniform: null argument");
    build_camera_uniform(*cam, viewport, out);
    return WS_OK;
}

int ws_build_settings_uniform(const ws_splatting_args* args, const ws_pointcloud* pc, ws_settings_uniform* out) {
    if (!args || !pc || !out) return fail(WS_ERR_INVALID, "ws_build_settings_uniform: null argument");
    std::memset(out, 0, sizeof *out);
    out->gaussian_scaling = args->gaussian_scaling;
    out->max_sh_deg = args->max_sh_deg;
    out->mip_splatting = args->has_mip_splatting ? (args->mip_splatting ? 1u : 0u) : ((pc->has_mip && pc->mip) ? 1u : 0u);
    out->kernel_size = args->has_kernel_size ? args->kernel_size : (pc->has_kernel_size ? pc->kernel_size : 0.3f);
    const ws_aabb& box = args->has_clipping_box ? args->clipping_box : pc->bbox;
    for (int i = 0; i < 3; ++i) {
        out->clip_min[i] = box.min[i];
        out->clip_max[i] = box.max[i];
        out->scene_center[i] = pc->center[i];  // always the point cloud centre (renderer.rs:644)
    }
    out->walltime = static_cast<float>(args->walltime_secs);  // Duration::as_secs_f32
    const float radius = ws_aabb_radius(&pc->bbox);
    const float extend = args->has_scene_extend ? args->scene_extend : radius;
    out->scene_extend = std::fmax(extend, radius);
    return WS_OK;
}

int ws_ply_rows_convert(const float* rows, uint32_t n, uint32_t sh_deg, void* gaussians_out, void* sh_out) {
    if ((!rows || !gaussians_out || !sh_out) && n) return fail(WS_ERR_INVALID, "ws_ply_rows_convert: null argument");
    if (sh_deg > 3) return fail(WS_ERR_UNSUPPORTED, "ws_ply_rows_convert: sh_deg > 3");
    const uint32_t num_coefs = (sh_deg + 1) * (sh_deg + 1);
    const size_t row_len = 3 + 3 + 3 + (num_coefs - 1) * 3 + 1 + 3 + 4;
    uint8_t* gout = static_cast<uint8_t*>(gaussians_out);
    uint8_t* sout = static_cast<uint8_t*>(sh_out);
    const ws::OmpQuietWorkers omp_quiet;  // (ws_internal.h: the region's workers sleep at once instead of spinning 200 ms)
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < static_cast<int64_t>(n); ++i) {
        const float* row = rows + static_cast<size_t>(i) * row_len;
        const float* rest = row + 9;
        const float* tail = rest + (num_coefs - 1) * 3;
        float sh[16][3];
        std::memset(sh, 0, sizeof sh);
        sh[0][0] = row[6];
        sh[0][1] = row[7];
        sh[0][2] = row[8];
        for (uint32_t c = 0; c + 1 < num_coefs; ++c)  // channel-first [3][C-1] -> [C][3]
            for (uint32_t j = 0; j < 3; ++j) sh[c + 1][j] = rest[j * (num_coefs - 1) + c];
        const float opacity = sigmoid(tail[0]);
        const float scale[3] = {std::exp(tail[1]), std::exp(tail[2]), std::exp(tail[3])};
        float q[4] = {tail[4], tail[5], tail[6], tail[7]};
        const float mag = std::sqrt(q[0] * q[0] + (q[1] * q[1] + q[2] * q[2] + q[3] * q[3]));
        const float inv = 1.0f / mag;
        for (float& v : q) v = v * inv;
        float cov[6];
        build_cov(q, scale, cov);
        uint8_t* g = gout + static_cast<size_t>(i) * 28;
        std::memcpy(g, row, 12);
        const uint16_t op = host_f32_to_f16(opacity), zero = 0;
        std::memcpy(g + 12, &op, 2);
        std::memcpy(g + 14, &zero, 2);
        for (int k = 0; k < 6; ++k) {
            const uint16_t h = host_f32_to_f16(cov[k]);
            std::memcpy(g + 16 + 2 * k, &h, 2);
        }
        uint8_t* s = sout + static_cast<size_t>(i) * 96;
        for (int c = 0; c < 16; ++c)
            for (int j = 0; j < 3; ++j) {
                const uint16_t h = host_f32_to_f16(sh[c][j]);
                std::memcpy(s + (c * 3 + j) * 2, &h, 2);
            }
    }
    return WS_OK;
}

int ws_pointcloud_stats(const void* gaussians, uint32_t n, uint32_t stride, const ws_aabb* start, ws_aabb* bbox,
                        float center[3], int32_t* has_up, float up[3]) {
    if (!gaussians || !start || !bbox || !center || n == 0 || stride < 12)
        return fail(WS_ERR_INVALID, "ws_pointcloud_stats: bad argument");
    const uint8_t* base = static_cast<const uint8_t*>(gaussians);
    *bbox = *start;
    float sum[3] = {0, 0, 0};
    for (uint32_t i = 0; i < n; ++i) {
        float p[3];
        std::memcpy(p, base + static_cast<size_t>(i) * stride, 12);
        for (int k = 0; k < 3; ++k) {
            bbox->min[k] = std::fmin(bbox->min[k], p[k]);
            bbox->max[k] = std::fmax(bbox->max[k], p[k]);
            sum[k] = sum[k] + p[k];
        }
    }
    const float inv_n = 1.0f / static_cast<float>(n);
    for (int k = 0; k < 3; ++k) center[k] = sum[k] * inv_n;
    bool ok = false;
    float nrm[3] = {0, 0, 0};
    if (n >= 3) {
        float xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
        for (uint32_t i = 0; i < n; ++i) {
            float p[3];
            std::memcpy(p, base + static_cast<size_t>(i) * stride, 12);
            const float rx = p[0] - center[0], ry = p[1] - center[1], rz = p[2] - center[2];
            xx += rx * rx;
            xy += rx * ry;
            xz += rx * rz;
            yy += ry * ry;
            yz += ry * rz;
            zz += rz * rz;
        }
        const float fn = static_cast<float>(n);
        xx /= fn;
        xy /= fn;
        xz /= fn;
        yy /= fn;
        yz /= fn;
        zz /= fn;
        float wd[3] = {0, 0, 0};
        const float axes[3][3] = {{yy * zz - yz * yz, xz * yz - xy * zz, xy * yz - xz * yy},
                                  {xz * yz - xy * zz, xx * zz - xz * xz, xy * xz - yz * xx},
                                  {xy * yz - xz * yy, xy * xz - yz * xx, xx * yy - xy * xy}};
        for (int a = 0; a < 3; ++a) {
            const float det = axes[a][a];
            float weight = det * det;
            if (wd[0] * axes[a][0] + wd[1] * axes[a][1] + wd[2] * axes[a][2] < 0.0f) weight = -weight;
            for (int k = 0; k < 3; ++k) wd[k] += axes[a][k] * weight;
        }
        const float mag = std::sqrt(wd[0] * wd[0] + wd[1] * wd[1] + wd[2] * wd[2]);
        const float inv = 1.0f / mag;
        for (int k = 0; k < 3; ++k) nrm[k] = wd[k] * inv;
        if (nrm[1] < 0.0f)
            for (float& v : nrm) v = -v;
        ok = std::isfinite(nrm[0]) && std::isfinite(nrm[1]) && std::isfinite(nrm[2]);
    }
    if (ws_aabb_radius(bbox) < 10.0f) ok = false;  // io/mod.rs:88-90
    if (has_up) *has_up = ok ? 1 : 0;
    if (up)
        for (int k = 0; k < 3; ++k) up[k] = ok ? nrm[k] : 0.0f;
    return WS_OK;
}

}  // extern "C"
