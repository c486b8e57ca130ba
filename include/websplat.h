/*
 * websplat.h -- C ABI of libwebsplat_hip.so, the MI355X (gfx950) drop-in for
 * web-splat's render hot path (GaussianRenderer::prepare + render and the
 * GPURSSorter they drive).
 *
 * Every entry point names the reference interface it replaces (file:line under
 * /root/reference).  Plain pointers and sizes only; no C++ / torch types.
 * All functions return WS_OK (0) or a negative ws_status; the message of the
 * last failure on the calling thread is available from ws_last_error().
 * Nothing throws across this boundary.
 *
 * Threading: handles are not thread-safe; use one context / renderer per host
 * thread (the reference records single-threaded, renderer.rs:191-260).
 * Async: prepare/render/sort only ENQUEUE work on the given HIP stream
 * (hipStream_t passed as void*; NULL = the default stream) and return; the
 * caller observes completion with ws_sync() -- the analogue of
 * queue.submit + device.poll(Wait) (bin/measure.rs:147).
 */
#ifndef WEBSPLAT_H
#define WEBSPLAT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WS_ABI_VERSION 3 /* 3: ws_context_config / ws_context_create_with_config; the library reads no environment variable */

typedef enum ws_status {
    WS_OK = 0,
    WS_ERR_INVALID = -1,     /* bad argument / layout mismatch */
    WS_ERR_HIP = -2,         /* a HIP runtime call failed (text in ws_last_error) */
    WS_ERR_OOM = -3,
    WS_ERR_UNSUPPORTED = -4, /* e.g. sh_deg > 3 */
    WS_ERR_STATE = -5,       /* e.g. render() before prepare() */
    WS_ERR_IO = -6,          /* loaders */
    WS_ERR_OVERFLOW = -7     /* device-side capacity exceeded (tile entries) */
} ws_status;

/* wgpu::TextureFormat choices the reference's front-ends use for the splat target:
 * Rgba8Unorm (lib.rs / bin/measure.rs), Rgba16Float (bin/render.rs:140), Rgba32Float (bin/video.rs). */
typedef enum ws_color_format {
    WS_FORMAT_RGBA8_UNORM = 0,
    WS_FORMAT_RGBA16_FLOAT = 1,
    WS_FORMAT_RGBA32_FLOAT = 2
} ws_color_format;

typedef struct ws_context ws_context;       /* wgpu Device+Queue      (lib.rs:57-125 WGPUContext) */
typedef struct ws_pointcloud ws_pointcloud; /* pointcloud.rs:72-88    PointCloud */
typedef struct ws_renderer ws_renderer;     /* renderer.rs:17-30      GaussianRenderer */
typedef struct ws_sorter ws_sorter;         /* gpu_rs.rs:23-42        GPURSSorter + PointCloudSortStuff */

/* pointcloud.rs:398-403 Aabb<f32> */
typedef struct ws_aabb {
    float min[3];
    float max[3];
} ws_aabb;

/* pointcloud.rs:360-396 Quantization / GaussianQuantization (64 B, uniform layout) */
typedef struct ws_quantization {
    int32_t zero_point;
    float scale;
    uint32_t _pad[2];
} ws_quantization;
typedef struct ws_gaussian_quantization {
    ws_quantization color_dc, color_rest, opacity, scaling_factor;
} ws_gaussian_quantization;

/* What io/mod.rs:27-43 GenericGaussianPointCloud hands to PointCloud::new (pointcloud.rs:99-199):
 * the loader's byte blobs, verbatim, in HOST memory.
 *   uncompressed: gaussians = num_points x 28 B  (pointcloud.rs:38-45 Gaussian)
 *                 sh_coefs  = num_points x 96 B  ([[f16;3];16], io/mod.rs:65)
 *   compressed:   gaussians = num_points x 24 B  (pointcloud.rs:14-22 GaussianCompressed)
 *                 sh_coefs  = packed int8, 3*(sh_deg+1)^2 B per SH entry (io/npz.rs:183-196)
 *                 covars    = n_covars x 12 B (pointcloud.rs:61-63 Covariance3D)
 *                 quantization = 64 B block */
typedef struct ws_pointcloud_desc {
    uint32_t num_points;
    uint32_t sh_deg;
    int32_t compressed;
    const void* gaussians;
    size_t gaussians_bytes;
    const void* sh_coefs;
    size_t sh_coefs_bytes;
    const void* covars; /* compressed only */
    size_t covars_bytes;
    const ws_gaussian_quantization* quantization; /* compressed only */
    ws_aabb bbox;
    float center[3];
    int32_t has_up;
    float up[3];
    int32_t has_mip_splatting;
    int32_t mip_splatting;
    int32_t has_kernel_size;
    float kernel_size;
    int32_t has_background_color;
    float background_color[3];
} ws_pointcloud_desc;

/* camera.rs:6-11 PerspectiveCamera + camera.rs:85-94 PerspectiveProjection */
typedef struct ws_camera {
    float position[3];
    float rotation[4]; /* cgmath Quaternion (s, x, y, z) */
    float fovx, fovy;  /* radians */
    float znear, zfar;
    float fov2view_ratio;
} ws_camera;

/* renderer.rs:585-599 SplattingArgs; Option<T> fields carry explicit has_* flags */
typedef struct ws_splatting_args {
    ws_camera camera;
    uint32_t viewport[2];
    float gaussian_scaling;
    uint32_t max_sh_deg;
    int32_t has_mip_splatting;
    int32_t mip_splatting;
    int32_t has_kernel_size;
    float kernel_size;
    int32_t has_clipping_box;
    ws_aabb clipping_box;
    double walltime_secs; /* Duration */
    int32_t has_scene_center;
    float scene_center[3]; /* ignored, like the reference (renderer.rs:644) */
    int32_t has_scene_extend;
    float scene_extend;
    double background_color[4]; /* wgpu::Color; used by callers for the clear only */
} ws_splatting_args;

/* renderer.rs:290-306 CameraUniform (272 B) and renderer.rs:602-618 SplattingArgsUniform (80 B) */
typedef struct ws_camera_uniform {
    float view[16], view_inv[16], proj[16], proj_inv[16];
    float viewport[2], focal[2];
} ws_camera_uniform;
typedef struct ws_settings_uniform {
    float clip_min[4], clip_max[4];
    float gaussian_scaling;
    uint32_t max_sh_deg;
    uint32_t mip_splatting;
    float kernel_size;
    float walltime;
    float scene_extend;
    uint32_t _pad[2];
    float scene_center[4];
} ws_settings_uniform;

/* per-stage GPU time, the reference's GPUStopwatch labels (renderer.rs:221,230; lib.rs:448) plus binning */
typedef struct ws_stage_times {
    float preprocess_ms;
    float sorting_ms;
    float binning_ms;
    float rasterization_ms;
} ws_stage_times;

/* one kernel launch of the last frame, in launch order (utils.rs:26-134 GPUStopwatch at kernel granularity) */
typedef struct ws_kernel_time {
    char name[40];
    float ms;
} ws_kernel_time;

/* device-side statistics of the last prepared frame (forces a sync) */
typedef struct ws_frame_stats {
    uint32_t num_visible;      /* V: renderer.rs:170-189 num_visible_points */
    uint32_t num_tile_entries; /* D: sum over visible splats of tiles touched */
    uint32_t tile_entries_capacity;
    uint32_t overflow;         /* 1 if D exceeded the capacity (entries dropped) */
} ws_frame_stats;

const char* ws_last_error(void);
uint32_t ws_abi_version(void);
/* bit 0: this is the EXPERIMENTAL build (make experimental, lib_exp/): it also carries the measured-and-lost variants the
 * exp_* fields of ws_context_config select; the product library refuses them with WS_ERR_UNSUPPORTED */
#define WS_BUILD_EXPERIMENTAL 1u
uint32_t ws_build_flags(void);

/* ---- context: lib.rs:68-125 WGPUContext::new_instance / new --------------------------------- */
/* ws_context_create uses the defaults of ws_context_config_init.  THE LIBRARY READS NO ENVIRONMENT VARIABLE: a drop-in
 * library must not be steered by the environment of whoever loads it.  Tuning / analysis switches travel in this struct;
 * bench.py, the tests and the tools/ mains translate their WS_* environment into it OUTSIDE the library
 * (include/websplat_env.h, web-splat_amd/websplat/api.py config_from_env). */
typedef struct ws_context_config {
    uint32_t struct_size;      /* sizeof(ws_context_config) of the caller: the struct may grow at its end */
    int32_t use_graph;         /* 0; 1 = prepare() on a real stream replays a captured frame graph (opt-in: ROCm 7.2 fault) */
    int32_t depth_skip_top;    /* 1; 0 = the depth sort always runs all of its passes (A/B) */
    int32_t blend_order;       /* -1 automatic; 0 image order; 1 longest list first; 2 / 3 measured experiments */
    int32_t blend_split;       /* -1 automatic (tiles < 2 x CUs); 0 / 1 never / always two half-tile workgroups per binning tile */
    int32_t bin_request;       /* 1 = decided per frame on the device (default); 0 never / 2 always bin at twice the blend's tile */
    int32_t batch_threads;     /* -1 automatic; 0 / 1 = a view batch never / always enqueues every slot from its own host thread */
    int32_t batch_queue_depth; /* -1 = default (5): frames a slot's host side may run ahead of the device; 0 = unbounded */
    int32_t blend_tpw_log2;    /* -1 automatic; tiles per blend workgroup = 2^n (4K-class tile counts) */
    int32_t blend_lds_pad_kb;  /* 0; unused dynamic LDS per blend workgroup (occupancy experiments) */
    int32_t tile_qw, tile_qh;  /* 4, 4: 8x8-px quadrants per compositing tile (2x2, 4x2 or 4x4) */
    int32_t debug_cut;         /* 0; analysis: stop every frame after stage n (1 = K1 ... 4 = tile sort) */
    int32_t capture;           /* 0; 1 = renderers keep per-splat source indices / per-tile debug words (tests) */
    int32_t render_views_fast_blend; /* 0; 1 = ws_render_views composites with the throughput blend instead of target precision */
    int32_t ply_decode_host;   /* 0; 1 = ws_load_ply converts the vertex rows on the host instead of on the device */
    int32_t depth_digit_bits;  /* 0 = default; 8 = four 8-bit passes (the reference's shape), 9 = three 9-bit passes over key - base */
    int32_t depth_tile_kpt;    /* 0 = by input size; 4 / 8 = keys per thread of the 9-bit depth sort's tiles (A/B) */
    int32_t blend_async;       /* -1 / 0 = k_blend (two workgroup barriers per staged batch); 1 = k_blend2 (double-buffered staging, LDS
                                * arrival counters, no per-batch barrier: bit-identical images, measured slower -- experimental build only) */
    /* measured-and-lost variants: honoured by the EXPERIMENTAL build only (lib_exp); the product library refuses non-defaults */
    int32_t exp_depth_sort;    /* 0 scan (default) | 1 fat-tile one-sweep | 2 one cooperative launch */
    int32_t exp_dsort_fat_grid;
    int32_t exp_blend_variant; /* 1 = k_blend_q */
    int32_t exp_blend_dma;     /* 1 = LDS-DMA staging */
    int32_t exp_batch_k1;      /* 1 (default) .. 4 views per K1 launch */
    int32_t exp_footprint_ellipse;
    int32_t exp_tile_sort_wide;
    int32_t reserved[6];       /* zero */
} ws_context_config;
void ws_context_config_init(ws_context_config* cfg); /* fills in the defaults above */
int ws_context_create(int hip_device, ws_context** out);
int ws_context_create_with_config(int hip_device, const ws_context_config* cfg, ws_context** out);
void ws_context_destroy(ws_context* ctx);
int ws_sync(ws_context* ctx, void* stream); /* device.poll(Wait) */
/* How a host thread of this process waits for the context's device in ws_sync / hipStreamSynchronize / the read-backs:
 * WS_HOST_WAIT_BLOCK sleeps until the completion interrupt -- what device.poll(wgpu::PollType::Wait) does behind
 * queue.submit (bin/measure.rs:147) -- WS_HOST_WAIT_SPIN polls (the HIP runtime's default: lowest wake-up latency, one
 * host core busy for as long as the wait lasts).  A process-wide property of the HIP device (hipSetDeviceFlags): eight
 * ranks of one node that wait spinning keep eight cores busy doing nothing. */
typedef enum ws_host_wait { WS_HOST_WAIT_SPIN = 0, WS_HOST_WAIT_BLOCK = 1 } ws_host_wait;
int ws_context_set_host_wait(ws_context* ctx, ws_host_wait mode);
int ws_device_info(ws_context* ctx, char* name, size_t name_len, uint32_t* num_cus, uint64_t* hbm_bytes);
/* plain device buffers for callers without their own allocator (tests, C++ drivers);
 * the analogue of device.create_buffer + queue.write_buffer + DownloadBuffer */
int ws_device_malloc(ws_context* ctx, size_t bytes, void** d_ptr);
int ws_device_free(ws_context* ctx, void* d_ptr);
int ws_memcpy_h2d(ws_context* ctx, void* d_dst, const void* h_src, size_t bytes, void* stream);
int ws_memcpy_d2h(ws_context* ctx, void* h_dst, const void* d_src, size_t bytes, void* stream);

/* ---- host-side boundary math (no GPU needed) ------------------------------------------------- */
/* camera.rs:26-35 PerspectiveCamera::fit_near_far */
int ws_camera_fit_near_far(ws_camera* cam, const ws_aabb* bbox);
/* scene.rs:85-108 impl Into<PerspectiveCamera> for SceneCamera (rotation = 3 rows of 3, as in cameras.json) */
int ws_camera_from_scene(const float position[3], const float rotation[9], float fx, float fy, uint32_t width,
                         uint32_t height, ws_camera* out);
/* renderer.rs:136-141 + 321-343 CameraUniform::set_camera / set_viewport / set_focal */
int ws_build_camera_uniform(const ws_camera* cam, const uint32_t viewport[2], ws_camera_uniform* out);
/* renderer.rs:620-651 SplattingArgsUniform::from_args_and_pc */
int ws_build_settings_uniform(const ws_splatting_args* args, const ws_pointcloud* pc, ws_settings_uniform* out);
/* pointcloud.rs:444-452 Aabb::center / radius */
float ws_aabb_radius(const ws_aabb* b);

/* ---- loaders' data prep (io/ply.rs:50-100, io/mod.rs:63-105, utils.rs:194-212) ---------------- */
/* Convert INRIA-layout PLY vertex rows (f32, little-endian, property order of io/ply.rs:54-88;
 * row length 3+3+3*(sh_deg+1)^2+1+3+4) into Gaussian (28 B) + SH (96 B) records. */
int ws_ply_rows_convert(const float* rows, uint32_t n, uint32_t sh_deg, void* gaussians_out, void* sh_out);
/* bbox (grown from `start`: Aabb::zeroed() for PLY io/mod.rs:74, Aabb::unit() for NPZ io/mod.rs:119),
 * centroid and plane-fit up vector.  stride = 28 or 24. */
int ws_pointcloud_stats(const void* gaussians, uint32_t n, uint32_t stride, const ws_aabb* start, ws_aabb* bbox,
                        float center[3], int32_t* has_up, float up[3]);
/* io/mod.rs:45-61 GenericGaussianPointCloud::load for a binary PLY file, then PointCloud::new */
int ws_pointcloud_load_ply(ws_context* ctx, const char* path, ws_pointcloud** out);
/* io/ply.rs:28-196 PlyReader::read + GenericGaussianPointCloud::new (io/mod.rs:63-105) on the HOST: the INRIA 3DGS
 * .ply decoded into the loader's byte blobs (Gaussian 28 B x N, SH 96 B x N; host memory owned by the returned
 * object), bbox grown from Aabb::zeroed(), centroid, plane fit, header comments.  No GPU needed. */
typedef struct ws_ply_cloud {
    uint32_t num_points;
    uint32_t sh_deg;
    const void* gaussians;
    size_t gaussians_bytes;
    const void* sh_coefs;
    size_t sh_coefs_bytes;
    ws_aabb bbox;
    float center[3];
    int32_t has_up;
    float up[3];
    int32_t has_mip_splatting;
    int32_t mip_splatting;
    int32_t has_kernel_size;
    float kernel_size;
    int32_t has_background_color;
    float background_color[3];
} ws_ply_cloud;
int ws_ply_read(const char* path, ws_ply_cloud** out);
void ws_ply_free(ws_ply_cloud* pc);

/* io/npz.rs:59-225 NpzReader::read: a c3dgs .npz decoded into the loader's byte blobs (HOST memory, owned by the
 * returned object): GaussianCompressed 24 B x N, packed int8 SH records 3*(sh_deg+1)^2 B, Covariance3D 12 B x M,
 * the 64-B quantisation block and the optional scalars.  No GPU needed. */
typedef struct ws_npz_cloud {
    uint32_t num_points;
    uint32_t sh_deg;
    const void* gaussians;
    size_t gaussians_bytes;
    const void* sh_coefs;
    size_t sh_coefs_bytes;
    const void* covars;
    size_t covars_bytes;
    ws_gaussian_quantization quantization;
    int32_t has_kernel_size;
    float kernel_size;
    int32_t has_mip_splatting;
    int32_t mip_splatting;
    int32_t has_background_color;
    float background_color[3];
} ws_npz_cloud;
int ws_npz_read(const char* path, ws_npz_cloud** out);
void ws_npz_free(ws_npz_cloud* pc);
/* NpzReader::read + GenericGaussianPointCloud::new_compressed (io/mod.rs:107-150) + PointCloud::new */
int ws_pointcloud_load_npz(ws_context* ctx, const char* path, ws_pointcloud** out);
/* io/mod.rs:45-61 GenericGaussianPointCloud::load: reader chosen by magic bytes ("ply" / "PK\3\4") */
int ws_pointcloud_load(ws_context* ctx, const char* path, ws_pointcloud** out);

/* ---- PointCloud: pointcloud.rs:99-222, 336-349 ------------------------------------------------ */
int ws_pointcloud_create(ws_context* ctx, const ws_pointcloud_desc* desc, ws_pointcloud** out);
/* PlyReader::read (io/ply.rs:50-100, 164-196) + GenericGaussianPointCloud::new (io/mod.rs:63-105) + PointCloud::new
 * with the per-vertex conversion on the GPU: `rows` = n raw vertex rows of the INRIA layout (14 + 3*(sh_deg+1)^2 f32
 * each, host memory, native endianness); bbox / centroid / up are computed from them on the host; `meta` (may be
 * NULL) supplies the optional mip_splatting / kernel_size / background_color header values (the other fields of the
 * descriptor are ignored).  ws_pointcloud_load_ply takes this route. */
int ws_pointcloud_create_from_ply_rows(ws_context* ctx, const float* rows, uint32_t n, uint32_t sh_deg,
                                       const ws_pointcloud_desc* meta, ws_pointcloud** out);
/* The resident scene as loader blobs again (28-B Gaussians + 96-B SH records, or the 24-B compressed records):
 * accessor / parity tooling, the analogue of reading the PointCloud's buffers back (pointcloud.rs:201-222). */
int ws_pointcloud_download(const ws_pointcloud* pc, void* gaussians, size_t gaussians_bytes, void* sh_coefs, size_t sh_coefs_bytes);
void ws_pointcloud_destroy(ws_pointcloud* pc);
uint32_t ws_pointcloud_num_points(const ws_pointcloud* pc);
uint32_t ws_pointcloud_sh_deg(const ws_pointcloud* pc);
int ws_pointcloud_compressed(const ws_pointcloud* pc);
int ws_pointcloud_bbox(const ws_pointcloud* pc, ws_aabb* out);
int ws_pointcloud_center(const ws_pointcloud* pc, float out[3]);
int ws_pointcloud_up(const ws_pointcloud* pc, float out[3]);                   /* returns 1 if Some */
int ws_pointcloud_mip_splatting(const ws_pointcloud* pc, int32_t* out);        /* returns 1 if Some */
int ws_pointcloud_kernel_size(const ws_pointcloud* pc, float* out);            /* returns 1 if Some */
int ws_pointcloud_background_color(const ws_pointcloud* pc, float out[3]);     /* returns 1 if Some */

/* ---- GaussianRenderer: renderer.rs:33-123, 170-260, 281 --------------------------------------- */
/* GaussianRenderer::new(device, queue, color_format, sh_deg, compressed) */
int ws_renderer_create(ws_context* ctx, ws_color_format format, uint32_t sh_deg, int compressed, ws_renderer** out);
void ws_renderer_destroy(ws_renderer* r);
ws_color_format ws_renderer_color_format(const ws_renderer* r);
/* GaussianRenderer::prepare: reset counters -> preprocess (K1/K1c) -> depth radix sort -> tile binning.
 * (Re)allocates per-renderer scratch when pc.num_points or the viewport changes (renderer.rs:200-211). */
int ws_renderer_prepare(ws_renderer* r, const ws_pointcloud* pc, const ws_splatting_args* args, void* stream);
/* begin_render_pass(clear = background) + GaussianRenderer::render: composites the prepared frame into
 * d_rgba_out (device memory, viewport.y rows of row_pitch_bytes; texel = 4 x {u8 | f16 | f32} by format),
 * premultiplied RGBA over `background` (the clear colour, bin/render.rs:113-116). */
int ws_renderer_render(ws_renderer* r, const ws_pointcloud* pc, const float background[4], void* d_rgba_out,
                       size_t row_pitch_bytes, void* stream);
/* GaussianRenderer::num_visible_points (syncs) */
int ws_renderer_num_visible(ws_renderer* r, uint32_t* out);
int ws_renderer_frame_stats(ws_renderer* r, ws_frame_stats* out); /* syncs */
/* Error bits of EVERY frame this renderer drew since creation / the last reset (syncs; bit 4 = a compositing workgroup waited
 * ~1 s for a staged batch that never came -- k_blend2's bounded spin): bit 0 = the (tile, splat)
 * entry list overflowed its capacity (entries_needed = what the last frame would have needed), bits 1..3 = a
 * look-back spin timed out.  The reference has no counterpart: wgpu validates sizes up front and the ROPs cannot
 * overflow; here the binned entry list can, and a caller that enqueues frames back to back (bin/measure.rs:98-153)
 * checks once after its wait. */
int ws_renderer_errors(ws_renderer* r, uint32_t* bits, uint32_t* entries_needed, int reset);
/* GPUStopwatch::take_measurements for the last frame (syncs); needs ws_renderer_enable_timers(r,1) */
/* enable: 0 = off, 1 = the four stage labels, 2 = additionally one HIP event pair per kernel launch */
int ws_renderer_enable_timers(ws_renderer* r, int enable);
int ws_renderer_stage_times(ws_renderer* r, ws_stage_times* out);
/* per-launch GPU time of the last frame (prepare + render), launch order; *count = launches recorded. Syncs.
 * Each time is a HIP-event interval = dispatch latency of a dependent launch + the kernel; the entry
 * "_empty_launch" (after the preprocess kernel) is an empty kernel recorded the same way (subtract it to compare with rocprofv3 durations). */
int ws_renderer_kernel_times(ws_renderer* r, uint32_t capacity, ws_kernel_time* out, uint32_t* count);
/* How render() composites (src/renderer.rs:63-67 PREMULTIPLIED_ALPHA_BLENDING on the pass's target):
 *   WS_BLEND_FAST (default)      front to back with early termination, accumulators in f32, ONE rounding at the store;
 *   WS_BLEND_TARGET_PRECISION    the reference's fixed-function blend literally: back to front over the clear colour,
 *                                the destination rounded to the target's precision (f16 RNE / unorm8 RNE / f32) after
 *                                EVERY splat, no early termination -- what bin/render.rs:154 (Rgba16Float) and
 *                                bin/measure.rs:184 (Rgba8Unorm) write.  Several times slower; ws_render_views uses it.
 *   WS_BLEND_FAST_EXACT_CUT      WS_BLEND_FAST, but a fragment within a few ulp of the cut-off (gaussian.wgsl:61, a > 2 CUTOFF
 *                                discards: a step of 0.009 * alpha in its weight) is kept or discarded by the reference's own
 *                                expression, dot(screen_pos, screen_pos) from the un-prescaled inverse, re-derived from the Splat
 *                                record (rare: ~1e-5 of the fragments).  No cut-off boundary pixel is left between this mode and
 *                                the f32 reference image (max-abs 6.1e-5, the early-out bound, on the uncompressed workloads);
 *                                the band test costs the blend +5 ... +7 % (frames/s -3 %), so it is a mode, not the default.
 * Takes effect at the next render(). */
typedef enum ws_blend_mode { WS_BLEND_FAST = 0, WS_BLEND_TARGET_PRECISION = 1, WS_BLEND_FAST_EXACT_CUT = 2 } ws_blend_mode;
int ws_renderer_set_blend_mode(ws_renderer* r, int mode);
/* parity tooling: also record the original Gaussian index of every store slot (costs 4 B per visible splat) */
int ws_renderer_enable_capture(ws_renderer* r, int enable);
/* Capacity of the (tile, splat) entry list; 0 = automatic: max(8 M, 4 per Gaussian per Mpixel), twice what the BASELINE
 * scenes need.  A frame that needs more sets error bit 0 and leaves its demand in a word that survives the per-frame reset;
 * once ws_renderer_errors (or ws_view_batch_errors) has read it, the next prepare() with the automatic capacity allocates
 * 1.25 x that demand.  Takes effect at the next prepare. */
int ws_renderer_set_tile_entry_capacity(ws_renderer* r, uint64_t entries);
/* parity read-back of the prepared frame (the reference's test tooling reads buffers back the same way,
 * gpu_rs.rs:900-941 download_buffer): splats = V x 20 B in store order, keys/src_index = V u32 in store
 * order (src_index = original Gaussian index of each slot), sorted = V u32 store indices in draw order
 * (far -> near).  Any pointer may be NULL.  capacity = number of elements each array can hold. Syncs. */
int ws_renderer_download_frame(ws_renderer* r, uint32_t capacity, void* splats, uint32_t* keys,
                               uint32_t* src_index, uint32_t* sorted, uint32_t* num_visible);
/* The binning tile the LAST prepared frame used: the context's tile (ws_context_tile_size) or, when the frame's splats span
 * several tiles, 2 x 2 blocks of it -- decided per frame on the device from the tile counts K1 sums for both sizes (a pure
 * function of the frame; WS_BIN_SHIFT=0 / 1 forces it off / on; frames in capture mode always use the context's tile).
 * Four compositing workgroups then share one binned list: half the (tile, splat) entries to emit and sort.  Syncs. */
int ws_renderer_binning_tile(ws_renderer* r, uint32_t* width, uint32_t* height);
/* Digit passes the depth sort of the LAST prepared frame executed, and (digit_bits, may be NULL) their width.  The reference
 * always runs four 8-bit passes (gpu_rs.rs:865-884).  Here the sort's first histogram kernel -- which reads every key anyway --
 * leaves the frame's key range on the device; the passes behind it take their digits from (key - base), and the last of the
 * four enqueued passes leaves at once when it would run over a constant digit (the identity): 3 passes on a frame whose keys
 * span less than 2^24 (8-bit digits: a camera outside the scene) or 2^27 (9-bit digits).  The compressed shader's keys
 * (preprocess_compressed.wgsl:325) are NOT confined to 24 bits: clip z is below znear for the nearest splats.  The digit
 * width is chosen per frame on the host from the key range the renderer's PREVIOUS frame posted (8 bits when it was below
 * 2^24, else 9; ws_context_config::depth_digit_bits forces it); either width gives the reference's stable order.  Syncs. */
int ws_renderer_depth_sort_passes(ws_renderer* r, uint32_t* passes);
int ws_renderer_depth_sort_digit_bits(ws_renderer* r, uint32_t* digit_bits); /* of the last prepared frame; no sync */
/* Analysis of FRAMES IN FLIGHT (no counterpart in the reference; rocprofv3's kernel trace serialises the hardware queues, so what
 * runs beside what has to be measured on the device): K1 and the compositing kernel of the next `frames` frames of this renderer
 * leave {first workgroup start, last workgroup end} on the device's 100-MHz clock -- stamps[frame][4] = K1 start, K1 end, blend
 * start, blend end; one clock shared by all renderers of the device.  frames = 0 switches it off.  download syncs. */
int ws_renderer_enable_frame_trace(ws_renderer* r, uint32_t frames);
int ws_renderer_download_frame_trace(ws_renderer* r, uint32_t capacity, uint64_t* stamps, uint32_t* count);
/* The compositing tile in pixels (one workgroup of the blend): 32x32 by default (four 16x16 tiles -- 4x4 wave quadrants
 * of 8x8 pixels -- sharing one binned list), 32x16 or 16x16 with WS_TILE_SHAPE=4x2|2x2 at context creation (tuning; 2x2 is
 * the literal one-workgroup-per-16x16-tile form).  Lists are built per BINNING tile: this tile, or 2 x 2 of them when the
 * frame decides so (ws_renderer_binning_tile). */
int ws_context_tile_size(const ws_context* ctx, uint32_t* width, uint32_t* height);
/* test hook, host only (no device work): the compositing pass's staging step for ONE (tile, splat) entry --
 * splat = the five 32-bit words of a 20-B Splat record (pointcloud.rs:352-358), tile origin in pixels ->
 * rec[10] = {i00, i01, c0, i10, i11, c1, alpha, r, g, b} (tile-local affine form of gaussian.wgsl:59-61 in the
 * exp2 domain) and the mask of 8x8-pixel quadrants (bit qy * (tile_w / 8) + qx) the kept ellipse may reach. */
int ws_debug_stage_splat(const uint32_t splat[5], float viewport_w, float viewport_h, float tile_x0, float tile_y0,
                         uint32_t tile_w, uint32_t tile_h, float rec[10], uint32_t* quadrant_mask);
/* test hook, host only: the binning footprint of ONE splat (words 0..2 of its 20-B record: v1, v2, pos) -- the ids
 * (ty * ceil(viewport_w / tile_w) + tx) of the binning tiles its kept ellipse a <= 2*CUTOFF (gaussian.wgsl:40-64) can
 * reach, in the order the binning stage emits them; *count = their number (what K1 stores per splat), of which at most
 * `capacity` are written.  tile_w / tile_h: 16 or 32. */
int ws_debug_footprint(const uint32_t splat[3], float viewport_w, float viewport_h, uint32_t tile_w, uint32_t tile_h,
                       uint32_t capacity, uint32_t* tiles, uint32_t* count);
/* test hooks, host only: (1) the packed tile rectangle a splat carries through the depth sort (x0 | y0 << 8 | (w - 1) << 16 |
 * (h - 1) << 24 in compositing tiles, 0xFFFFFFFF = lists no tile): the number of tiles it lists at the compositing tile and
 * at 2 x 2 of them, and the same rectangle in units of 2 x 2 tiles -- the arithmetic K1, k_bin_prefix and k_bin_emit share;
 * (2) the frame's binning decision from K1's per-slot sums of those two counts (request: 0 = never coarse, 1 = decide,
 * 2 = always; nslots <= 16): *shift = 0 (lists per compositing tile) or 1 (per 2 x 2 of them). */
int ws_debug_packed_rect(uint32_t rect, uint32_t* tiles, uint32_t* tiles_coarse, uint32_t* rect_coarse);
int ws_debug_binning_decision(uint32_t request, const uint32_t* sums, const uint32_t* sums_coarse, uint32_t nslots,
                              uint32_t* shift);
/* host twin of the depth sort's range decision (ws_internal.h depth_range_decide; CPU unit test, not on any render path): from the
 * frame's smallest and largest depth key and the radix (256 | 512) -> the base the passes behind the first subtract, whether the
 * fourth pass is the identity (skip), and the span class the next frame's digit width is chosen by (0 unknown, 1 = < 2^24, 2 = not) */
int ws_debug_depth_range(uint32_t key_min, uint32_t key_max, int have_keys, uint32_t digits, uint32_t* base, uint32_t* skip,
                         uint32_t* span_class);
/* tuning / analysis read-back: per tile LIST (one per binning tile, ws_renderer_binning_tile; row-major over
 * ceil(viewport / binning tile)), the length of the depth-ordered splat list and (capture mode, where the binning tile is
 * the compositing tile) how deep into it the compositing pass read: the position, counted from the near end, of the deepest
 * entry any of the tile's waves composited before its pixels were saturated.  Syncs. */
int ws_renderer_download_tile_stats(ws_renderer* r, uint32_t capacity, uint32_t* list_len, uint32_t* consumed,
                                    uint32_t* num_tiles);
/* analysis read-back (capture mode): walked[t * 17 + w] = staged records wave w of tile t composited (w < waves
 * per tile, 16 at the default tile), walked[t * 17 + 16] = sum over the tile's batches of the most any of its waves
 * composited in that batch -- the lock-step cost of the per-batch barriers.  Syncs. */
int ws_renderer_download_wave_stats(ws_renderer* r, uint32_t tile_capacity, uint32_t* walked);
/* analysis / parity read-back of the compositing schedule: up to 4096 tiles (1080p class) the compositing workgroups of a
 * renderer that draws one frame at a time (not a slot of a view batch with several slots, and its context's last prepare()
 * calls all on one stream) run longest list first (one small kernel behind the tile-id sort orders them; the image does not depend on the order).
 * order4[4 * b + 0..3] = (tx | ty << 16, begin, end, 0) for workgroup b: the blend tile it composites and that tile's entry
 * range; 0xFFFFFFFF in word 0 = no tile.  *num_blocks = 0 when the last prepared frame was not ordered (4K-class tile
 * counts, non-default tile shapes, WS_BLEND_ORDER=0).  Syncs. */
int ws_renderer_download_blend_order(ws_renderer* r, uint32_t capacity_blocks, uint32_t* order4, uint32_t* num_blocks);
/* analysis: the next render() launches the time-stamped build of the compositing kernel (production form: 32x32 tiles,
 * rgba32float target, the frame's own binning) and every wave of every tile leaves 16 words: cycles (shader clock) spent in
 * [0] the tile-range load, [1] the first batch's dependent gather chain, [2] later batches' gather waits, [3] decode,
 * [4] the staging barrier, [5] compaction, [6] the walk, [7] the end-of-batch vote, [8] the pixel store; [9] batches,
 * [10] records walked, [11] / [12] shader-clock stamps at start / end, [13] / [14] the 100-MHz clock at start / end,
 * [15] XCC id << 28 | HW_ID.  The counterpart of the reference's GPUStopwatch (utils.rs:26-134) below kernel granularity.
 * times[(t * 16 + w) * 16 + k] for blend tile t (row-major), wave w.  Syncs. */
int ws_renderer_enable_blend_timing(ws_renderer* r, int enable);
int ws_renderer_download_blend_timing(ws_renderer* r, uint32_t tile_capacity, uint32_t* times, uint32_t* num_tiles);
/* parity read-back of the binning result: binning tile t's depth-ordered (far -> near) splat list is
 * entries[begin[t] .. end[t]) (store indices, as `sorted` of ws_renderer_download_frame); t is row-major over
 * ceil(viewport / binning tile) (ws_renderer_binning_tile; the tile count comes from ws_renderer_download_tile_stats).
 * Any pointer may be NULL; *num_entries = D.  Syncs. */
int ws_renderer_download_tile_lists(ws_renderer* r, uint32_t tile_capacity, uint32_t* begin, uint32_t* end,
                                    uint32_t entry_capacity, uint32_t* entries, uint32_t* num_entries);

/* ---- Scene: scene.rs:13-24, 113-194 (host only) ----------------------------------------------------- */
#define WS_SPLIT_ALL (-1)
#define WS_SPLIT_TRAIN 0 /* scene.rs:63-67 Split::Train */
#define WS_SPLIT_TEST 1  /* Split::Test: every 8th camera of the file (scene.rs:143-151) */
typedef struct ws_scene ws_scene; /* scene.rs:113-118 Scene */
/* scene.rs:13-24 SceneCamera; rotation = the 3 rows of cameras.json's 3x3 (camera-to-world) */
typedef struct ws_scene_camera {
    uint32_t id;
    char img_name[128];
    uint32_t width, height;
    float position[3];
    float rotation[9];
    float fx, fy;
    int32_t split;
} ws_scene_camera;
int ws_scene_load_json(const char* path, ws_scene** out);                        /* Scene::from_json */
int ws_scene_from_json_text(const char* text, size_t len, ws_scene** out);
void ws_scene_destroy(ws_scene* s);
uint32_t ws_scene_num_cameras(const ws_scene* s);
float ws_scene_extend(const ws_scene* s);                                        /* max camera-to-camera distance */
/* Scene::cameras(split), sorted by id; returns the number of matching cameras, fills at most `capacity` */
uint32_t ws_scene_cameras(const ws_scene* s, int split, uint32_t capacity, ws_scene_camera* out);
int ws_scene_get_camera(const ws_scene* s, uint32_t id, ws_scene_camera* out);   /* Scene::camera; 1 if Some */
int ws_scene_nearest_camera(const ws_scene* s, const float pos[3], int split, uint32_t* id); /* 1 if Some */

/* ---- offline front-ends: bin/render.rs, bin/measure.rs, renderer.rs:417-583 Display ------------------- */
/* bin/render.rs:187-246 download_texture: device image (format of the renderer) -> host RGBA8, each channel
 * clamp(v, 0, 1) * 255 TRUNCATED (`as u8`); unorm8 images are copied. out = width*height*4 bytes. Syncs. */
int ws_download_texture_rgba8(ws_context* ctx, const void* d_image, ws_color_format format, uint32_t width,
                              uint32_t height, size_t row_pitch_bytes, uint8_t* out, void* stream);
/* `image` crate save (bin/render.rs:127): RGBA8 PNG */
int ws_png_write_rgba8(const char* path, uint32_t width, uint32_t height, const uint8_t* rgba, size_t row_stride_bytes);
/* bin/render.rs:33-128 render_views: every camera of `split` (sorted by id) at its own resolution capped to 1600 px
 * wide (height rescaled with truncation), Rgba16Float target cleared to TRANSPARENT, fit_near_far, walltime 100 s,
 * max_sh_deg = pc.sh_deg  ->  <out_dir>/<train|test>/<index:05>.png.  *rendered = images written. */
int ws_render_views(ws_context* ctx, const ws_pointcloud* pc, const ws_scene* scene, int split, const char* out_dir,
                    uint32_t* rendered);
/* bin/measure.rs:27-154 render_views: 2048x2048 Rgba8Unorm target, one warm-up frame of camera 0, then num_samples
 * (reference: 10) frames of every TRAIN camera back to back, one sync; *fps = 1 / (elapsed / (cameras*num_samples))
 * with the clock started BEFORE the warm-up frame, as the reference does.  frames_in_flight > 1 (not in the
 * reference) gives every in-flight frame its own renderer scratch, target and HIP stream. */
int ws_measure(ws_context* ctx, const ws_pointcloud* pc, const ws_scene* scene, uint32_t num_samples,
               uint32_t frames_in_flight, float* fps);
/* ---- view batches (BASELINE configs 4 / 5: many independent views of one resident scene) ----------------
 * The reference renders one view at a time on one queue (lib.rs:422-431, bin/measure.rs:98-146).  A view batch keeps
 * `frames_in_flight` frames going at once: frame i of the batch's life runs on renderer + HIP stream i mod
 * frames_in_flight (private scratch each; the point cloud is shared).  ws_view_batch_render only ENQUEUES; the caller
 * observes completion with ws_view_batch_sync.  The host's RUN-AHEAD is bounded: a slot's host side stays at most 5 frames
 * (WS_BATCH_QUEUE_DEPTH; 0 = unbounded) ahead of the device, so a call with more views than slots x 5 returns when all but
 * the last of them have reached the device -- the caller's thread SLEEPS meanwhile (it polls a word the compositing kernel
 * posts to pinned memory; no runtime call, no spinning: a rank costs 0.5-0.7 host cores instead of 1.9).  On an error in the
 * middle of a call the one-thread path stops at the failing frame; with submission threads (below) the other slots still
 * enqueue THEIR frames of the call, and the frame-to-slot position advances by num_views.  For point clouds of at most 512 Ki Gaussians -- where the GPU needs less time
 * per frame than one host thread needs to enqueue it -- every slot's frames are enqueued by a worker thread of the batch (the
 * order on each stream is unchanged; the call returns when everything is enqueued; WS_BATCH_THREADS=0 / 1 forces it off / on).
 * d_targets[i] receives view i (device memory, format of the batch);
 * targets may repeat with period frames_in_flight (a ring), since a slot's frames are ordered on its stream. */
typedef struct ws_view_batch ws_view_batch;
int ws_view_batch_create(ws_context* ctx, ws_color_format format, uint32_t sh_deg, int compressed,
                         uint32_t frames_in_flight, ws_view_batch** out);
void ws_view_batch_destroy(ws_view_batch* b);
uint32_t ws_view_batch_frames_in_flight(const ws_view_batch* b);
int ws_view_batch_render(ws_view_batch* b, const ws_pointcloud* pc, const ws_splatting_args* views, uint32_t num_views,
                         void* const* d_targets, size_t row_pitch_bytes, const float background[4]);
int ws_view_batch_sync(ws_view_batch* b);
int ws_view_batch_errors(ws_view_batch* b, uint32_t* bits, int reset); /* OR of ws_renderer_errors over the slots (syncs) */
ws_renderer* ws_view_batch_renderer(ws_view_batch* b, uint32_t slot); /* the renderer of a slot (stats, timers) */
/* times ws_view_batch_render had to sleep because a slot's host side was queue_depth frames ahead of the device (statistics).
 * A slot whose progress word does not move for 10 s (lost launch, device fault) makes ws_view_batch_render return
 * WS_ERR_STATE once; its later frames are enqueued without the bound. */
uint32_t ws_view_batch_host_waits(const ws_view_batch* b);

/* Display::render (renderer.rs:548-582) + display.wgsl:37-55: the splat image (premultiplied RGBA, renderer
 * format) composited with PREMULTIPLIED_ALPHA_BLENDING over a surface cleared to `background`, written as 8-bit
 * unorm in the surface's channel order (lib.rs:184-243 picks the surface format and strips the sRGB suffix). */
typedef enum ws_surface_format { WS_SURFACE_RGBA8_UNORM = 0, WS_SURFACE_BGRA8_UNORM = 1 } ws_surface_format;
int ws_display_composite(ws_context* ctx, const void* d_src, ws_color_format src_format, size_t src_pitch_bytes,
                         uint32_t width, uint32_t height, const float background[4], ws_surface_format dst_format,
                         void* d_dst, size_t dst_pitch_bytes, void* stream);

/* ---- GPURSSorter: gpu_rs.rs:65-175, 720-727, 865-884 ------------------------------------------ */
/* GPURSSorter::new + create_sort_stuff(device, max_n): scratch for sorting up to max_n pairs */
int ws_sorter_create(ws_context* ctx, uint32_t max_n, ws_sorter** out);
void ws_sorter_destroy(ws_sorter* s);
/* record_sort (d_count == NULL, sorts n pairs) / record_sort_indirect (count read from device memory,
 * clamped to n): ascending, stable, in place in d_keys / d_payload (gpu_rs.rs:865-884). */
int ws_sorter_sort(ws_sorter* s, uint32_t* d_keys, uint32_t* d_payload, const uint32_t* d_count, uint32_t n,
                   void* stream);
/* The same contract (record_sort / record_sort_indirect) through the kernels a frame's depth sort runs: four 8-bit
 * passes of the generic sorter -- or, in a context created with WS_DEPTH_SORT=onesweep | coop, the fat-tile one-sweep --
 * with d_aux (may be NULL), a 4-byte companion that travels with the payload.  In place.
 * d_keys must be 16-byte aligned (both entry points: the histogram kernels read the keys four at a time);
 * WS_ERR_INVALID otherwise. */
int ws_sorter_sort_depth(ws_sorter* s, uint32_t* d_keys, uint32_t* d_payload, uint32_t* d_aux, const uint32_t* d_count,
                         uint32_t n, void* stream);
/* GPURSSorter::test_sort (gpu_rs.rs:295-331): 8192 reversed f32 keys must come out ascending. 1 = pass */
int ws_sort_selftest(ws_context* ctx, int* passed);

#ifdef __cplusplus
}
#endif
#endif /* WEBSPLAT_H */
