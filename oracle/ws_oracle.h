/*
 * ws_oracle.h -- CPU oracle for the web-splat render hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * PARITY STATUS.  The reference (KeKsBoTer/web-splat) ships no golden vectors,
 * fixtures or tests for K1 / K1c / K6 and cannot be built here (Rust + WGSL
 * through wgpu/naga; no cargo, no Vulkan ICD), so there is no oracle/_ref.
 *   PINNED to the reference's own source:
 *   - the sort contract, by GPURSSorter::test_sort (src/gpu_rs.rs:295-331: 8192
 *     reversed f32 keys), tests/test_oracle.py, and by radix_sort.wgsl executed
 *     from source as GPURSSorter::record_sort drives it (tests/golden/wgsl_sort_*.npz:
 *     ties, extreme bit patterns, one and two scatter blocks): ascending u32, stable;
 *   - wso_preprocess (K1), wso_preprocess_compressed (K1c) and the fragment
 *     function inside wso_render (K6), by tests/golden/wgsl_*.npz: the outputs of
 *     preprocess.wgsl / preprocess_compressed.wgsl / gaussian.wgsl executed FROM
 *     THEIR SOURCE TEXT by the WGSL-subset interpreter oracle/wgsl_exec.py in the
 *     build container (generator: tests/golden/gen_wgsl_golden.py).  K1 agrees
 *     bit for bit on all ten cases (tests/test_wgsl_golden.py).
 *   parity unpinned (line-by-line restatement only): the Rust host math
 *   (camera.rs, cgmath, renderer.rs uniforms), the rasteriser's interpolation and
 *   the blend state of the draw (renderer.rs:65), the loaders (ws_oracle_io.py).
 */
#ifndef WS_ORACLE_H
#define WS_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- IEEE binary16 (half crate 2.6.0 f16::from_f32 / to_f32; WGSL pack2x16float) */
uint16_t wso_f32_to_f16(float f);
float wso_f16_to_f32(uint16_t h);

/* ---- camera + uniforms (src/camera.rs, src/renderer.rs:290-343, 602-651) */
typedef struct {
    float position[3];
    float rotation[4]; /* quaternion (s, x, y, z) as cgmath::Quaternion::new */
    float fovx, fovy;  /* radians */
    float znear, zfar;
    float fov2view_ratio;
} wso_camera;

typedef struct {
    float min[3];
    float max[3];
} wso_aabb;

/* CameraUniform, 272 B: view, view_inv, proj (Y-flipped), proj_inv (of the
 * un-flipped proj), viewport, focal.  Matrices column-major. */
typedef struct {
    float view[16];
    float view_inv[16];
    float proj[16];
    float proj_inv[16];
    float viewport[2];
    float focal[2];
} wso_camera_uniform;

/* SplattingArgsUniform, 80 B (src/renderer.rs:602-618) */
typedef struct {
    float clip_min[4];
    float clip_max[4];
    float gaussian_scaling;
    uint32_t max_sh_deg;
    uint32_t mip_splatting;
    float kernel_size;
    float walltime;
    float scene_extend;
    uint32_t _pad[2];
    float scene_center[4];
} wso_settings_uniform;

/* {zero_point i32, scale f32, pad x2} x {color_dc, color_rest, opacity, scaling_factor} */
typedef struct {
    int32_t zero_point;
    float scale;
    uint32_t _pad[2];
} wso_quantization;
typedef struct {
    wso_quantization color_dc, color_rest, opacity, scaling_factor;
} wso_gaussian_quantization;

void wso_quat_to_mat3(const float q[4], float m9[9]);                       /* cgmath From<Quaternion> for Matrix3 */
void wso_mat3_to_quat(const float m9[9], float q[4]);                       /* cgmath From<Matrix3> for Quaternion */
void wso_world2view(const float r9[9], const float t[3], float out16[16]);  /* camera.rs:207-214 */
void wso_build_proj(float znear, float zfar, float fovx, float fovy, float out16[16]); /* camera.rs:216-234 */
float wso_fov2focal(float fov, float pixels);                               /* camera.rs:240-242 */
float wso_focal2fov(float focal, float pixels);                             /* camera.rs:236-238 */
void wso_fit_near_far(wso_camera* cam, const wso_aabb* bbox);               /* camera.rs:26-35 */
float wso_aabb_radius(const wso_aabb* b);                                   /* pointcloud.rs:449-452 */
void wso_camera_uniform_build(const wso_camera* cam, uint32_t vw, uint32_t vh, wso_camera_uniform* out);
/* scene.rs:85-108: cameras.json entry -> PerspectiveCamera */
void wso_scene_camera_to_perspective(const float position[3], const float rotation_rows[9],
                                     float fx, float fy, uint32_t width, uint32_t height,
                                     wso_camera* out);

/* ---- loader data prep (io/ply.rs:50-100, utils.rs:194-212, io/mod.rs:63-105) */
float wso_sigmoid(float x);
void wso_build_cov(const float q[4], const float scale[3], float out6[6]);
/* rows: n x (3+3+3*(sh_deg+1)^2+1+3+4) f32 in INRIA property order;
 * gaussians: n x 28 B; sh: n x 96 B */
void wso_ply_rows_convert(const float* rows, uint32_t n, uint32_t sh_deg, uint8_t* gaussians, uint8_t* sh);
/* bbox (starting from `start`), centroid, up vector; returns 1 if `up` valid */
int wso_pointcloud_stats(const uint8_t* gaussians, uint32_t n, uint32_t stride, const wso_aabb* start,
                         wso_aabb* bbox, float center[3], float up[3]);

/* ---- K1 / K1c : src/shaders/preprocess.wgsl:163-280, preprocess_compressed.wgsl:206-332
 * Deterministic compaction: store order == Gaussian index order.
 * splats: v x 20 B, keys: v, src_index[v] = original Gaussian index.  Returns v. */
uint32_t wso_preprocess(const uint8_t* gaussians, const uint8_t* sh, uint32_t n,
                        const wso_camera_uniform* cam, const wso_settings_uniform* rs,
                        uint8_t* splats, uint32_t* keys, uint32_t* src_index);
uint32_t wso_preprocess_compressed(const uint8_t* gaussians, const uint8_t* sh_bytes, const uint8_t* covars,
                                   const wso_gaussian_quantization* q, uint32_t n, uint32_t sh_deg_layout,
                                   const wso_camera_uniform* cam, const wso_settings_uniform* rs,
                                   uint8_t* splats, uint32_t* keys, uint32_t* src_index);

/* ---- sort contract: src/gpu_rs.rs:865-884 -- ascending, stable, (u32,u32) pairs; 4 x 8-bit LSD */
void wso_sort_pairs(uint32_t* keys, uint32_t* payload, uint32_t n);

/* ---- K6: src/shaders/gaussian.wgsl:30-67 + PREMULTIPLIED_ALPHA_BLENDING (renderer.rs:65)
 * target_mode: 0 = f32 target (video.rs), 1 = f16 rounding after every blend (render.rs),
 *              2 = unorm8 rounding after every blend (viewer / measure.rs).
 * out: h x w x 4 f32 premultiplied RGBA, cleared to `background` first. */
void wso_render(const uint8_t* splats, const uint32_t* sorted_indices, uint32_t v, uint32_t w, uint32_t h,
                const float background[4], int target_mode, float* out);

int wso_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
