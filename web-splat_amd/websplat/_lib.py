"""ctypes binding of libwebsplat_hip.so (include/websplat.h).

This is the Python-side stub a maintainer would write against the C ABI; it contains no
compute and no fallback: if the HIP library is missing, importing fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# WEBSPLAT_LIB selects another build of the SAME library (kernel tuning A/B runs); there is still no fallback.
LIB_PATH = os.environ.get("WEBSPLAT_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libwebsplat_hip.so")


class WebSplatError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"websplat error {code}: {msg}")
        self.code = code


WS_OK = 0
WS_ERR_INVALID = -1
WS_ERR_HIP = -2
WS_ERR_OOM = -3
WS_ERR_UNSUPPORTED = -4
WS_ERR_STATE = -5
WS_ERR_IO = -6
WS_ERR_OVERFLOW = -7

WS_FORMAT_RGBA8_UNORM = 0
WS_FORMAT_RGBA16_FLOAT = 1
WS_FORMAT_RGBA32_FLOAT = 2


class ws_aabb(C.Structure):
    _fields_ = [("min", C.c_float * 3), ("max", C.c_float * 3)]


class ws_quantization(C.Structure):
    _fields_ = [("zero_point", C.c_int32), ("scale", C.c_float), ("_pad", C.c_uint32 * 2)]


class ws_gaussian_quantization(C.Structure):
    _fields_ = [("color_dc", ws_quantization), ("color_rest", ws_quantization),
                ("opacity", ws_quantization), ("scaling_factor", ws_quantization)]


class ws_context_config(C.Structure):
    """include/websplat.h ws_context_config: every tuning / analysis switch of a context.  The library reads no environment
    variable; config_from_env() below is the harness-side translation of the historical WS_* variables."""
    _fields_ = [(n, C.c_uint32 if n == "struct_size" else C.c_int32) for n in (
        "struct_size", "use_graph", "depth_skip_top", "blend_order", "blend_split", "bin_request", "batch_threads",
        "batch_queue_depth", "blend_tpw_log2", "blend_lds_pad_kb", "tile_qw", "tile_qh", "debug_cut", "capture",
        "render_views_fast_blend", "ply_decode_host", "depth_digit_bits", "depth_tile_kpt", "blend_async", "exp_depth_sort", "exp_dsort_fat_grid",
        "exp_blend_variant", "exp_blend_dma", "exp_batch_k1", "exp_footprint_ellipse", "exp_tile_sort_wide")] + [("reserved", C.c_int32 * 6)]


class ws_pointcloud_desc(C.Structure):
    _fields_ = [
        ("num_points", C.c_uint32), ("sh_deg", C.c_uint32), ("compressed", C.c_int32),
        ("gaussians", C.c_void_p), ("gaussians_bytes", C.c_size_t),
        ("sh_coefs", C.c_void_p), ("sh_coefs_bytes", C.c_size_t),
        ("covars", C.c_void_p), ("covars_bytes", C.c_size_t),
        ("quantization", C.POINTER(ws_gaussian_quantization)),
        ("bbox", ws_aabb), ("center", C.c_float * 3),
        ("has_up", C.c_int32), ("up", C.c_float * 3),
        ("has_mip_splatting", C.c_int32), ("mip_splatting", C.c_int32),
        ("has_kernel_size", C.c_int32), ("kernel_size", C.c_float),
        ("has_background_color", C.c_int32), ("background_color", C.c_float * 3),
    ]


class ws_camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("rotation", C.c_float * 4),
                ("fovx", C.c_float), ("fovy", C.c_float), ("znear", C.c_float), ("zfar", C.c_float),
                ("fov2view_ratio", C.c_float)]


class ws_splatting_args(C.Structure):
    _fields_ = [
        ("camera", ws_camera), ("viewport", C.c_uint32 * 2),
        ("gaussian_scaling", C.c_float), ("max_sh_deg", C.c_uint32),
        ("has_mip_splatting", C.c_int32), ("mip_splatting", C.c_int32),
        ("has_kernel_size", C.c_int32), ("kernel_size", C.c_float),
        ("has_clipping_box", C.c_int32), ("clipping_box", ws_aabb),
        ("walltime_secs", C.c_double),
        ("has_scene_center", C.c_int32), ("scene_center", C.c_float * 3),
        ("has_scene_extend", C.c_int32), ("scene_extend", C.c_float),
        ("background_color", C.c_double * 4),
    ]


class ws_camera_uniform(C.Structure):
    _fields_ = [("view", C.c_float * 16), ("view_inv", C.c_float * 16), ("proj", C.c_float * 16),
                ("proj_inv", C.c_float * 16), ("viewport", C.c_float * 2), ("focal", C.c_float * 2)]


class ws_settings_uniform(C.Structure):
    _fields_ = [("clip_min", C.c_float * 4), ("clip_max", C.c_float * 4),
                ("gaussian_scaling", C.c_float), ("max_sh_deg", C.c_uint32), ("mip_splatting", C.c_uint32),
                ("kernel_size", C.c_float), ("walltime", C.c_float), ("scene_extend", C.c_float),
                ("_pad", C.c_uint32 * 2), ("scene_center", C.c_float * 4)]


class ws_stage_times(C.Structure):
    _fields_ = [("preprocess_ms", C.c_float), ("sorting_ms", C.c_float), ("binning_ms", C.c_float),
                ("rasterization_ms", C.c_float)]


class ws_npz_cloud(C.Structure):
    _fields_ = [("num_points", C.c_uint32), ("sh_deg", C.c_uint32),
                ("gaussians", C.c_void_p), ("gaussians_bytes", C.c_size_t),
                ("sh_coefs", C.c_void_p), ("sh_coefs_bytes", C.c_size_t),
                ("covars", C.c_void_p), ("covars_bytes", C.c_size_t),
                ("quantization", ws_gaussian_quantization),
                ("has_kernel_size", C.c_int32), ("kernel_size", C.c_float),
                ("has_mip_splatting", C.c_int32), ("mip_splatting", C.c_int32),
                ("has_background_color", C.c_int32), ("background_color", C.c_float * 3)]


class ws_ply_cloud(C.Structure):
    _fields_ = [("num_points", C.c_uint32), ("sh_deg", C.c_uint32),
                ("gaussians", C.c_void_p), ("gaussians_bytes", C.c_size_t),
                ("sh_coefs", C.c_void_p), ("sh_coefs_bytes", C.c_size_t),
                ("bbox", ws_aabb), ("center", C.c_float * 3), ("has_up", C.c_int32), ("up", C.c_float * 3),
                ("has_mip_splatting", C.c_int32), ("mip_splatting", C.c_int32),
                ("has_kernel_size", C.c_int32), ("kernel_size", C.c_float),
                ("has_background_color", C.c_int32), ("background_color", C.c_float * 3)]


class ws_scene_camera(C.Structure):
    _fields_ = [("id", C.c_uint32), ("img_name", C.c_char * 128), ("width", C.c_uint32), ("height", C.c_uint32),
                ("position", C.c_float * 3), ("rotation", C.c_float * 9), ("fx", C.c_float), ("fy", C.c_float),
                ("split", C.c_int32)]


WS_SPLIT_ALL, WS_SPLIT_TRAIN, WS_SPLIT_TEST = -1, 0, 1
WS_SURFACE_RGBA8_UNORM, WS_SURFACE_BGRA8_UNORM = 0, 1


class ws_kernel_time(C.Structure):
    _fields_ = [("name", C.c_char * 40), ("ms", C.c_float)]


class ws_frame_stats(C.Structure):
    _fields_ = [("num_visible", C.c_uint32), ("num_tile_entries", C.c_uint32),
                ("tile_entries_capacity", C.c_uint32), ("overflow", C.c_uint32)]


assert C.sizeof(ws_camera_uniform) == 272
assert C.sizeof(ws_settings_uniform) == 80
assert C.sizeof(ws_gaussian_quantization) == 64

_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
_u32p = C.POINTER(C.c_uint32)
_f32p = C.POINTER(C.c_float)

# name -> (restype, argtypes): every symbol include/websplat.h declares
SIGNATURES = {
    "ws_last_error": (C.c_char_p, []),
    "ws_abi_version": (C.c_uint32, []),
    "ws_build_flags": (C.c_uint32, []),
    "ws_context_create": (C.c_int, [C.c_int, _PP]),
    "ws_context_config_init": (None, [C.POINTER(ws_context_config)]),
    "ws_context_create_with_config": (C.c_int, [C.c_int, C.POINTER(ws_context_config), _PP]),
    "ws_context_destroy": (None, [_P]),
    "ws_context_tile_size": (C.c_int, [_P, _u32p, _u32p]),
    "ws_renderer_download_wave_stats": (C.c_int, [_P, C.c_uint32, _u32p]),
    "ws_renderer_download_blend_order": (C.c_int, [_P, C.c_uint32, _P, _P]),
    "ws_renderer_depth_sort_passes": (C.c_int, [_P, _P]),
    "ws_renderer_depth_sort_digit_bits": (C.c_int, [_P, _P]),
    "ws_renderer_enable_frame_trace": (C.c_int, [_P, C.c_uint32]),
    "ws_renderer_download_frame_trace": (C.c_int, [_P, C.c_uint32, _P, _P]),
    "ws_renderer_enable_blend_timing": (C.c_int, [_P, C.c_int]),
    "ws_renderer_download_blend_timing": (C.c_int, [_P, C.c_uint32, _P, _P]),
    "ws_debug_stage_splat": (C.c_int, [_u32p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint32,
                                       _f32p, _u32p]),
    "ws_debug_packed_rect": (C.c_int, [C.c_uint32, _u32p, _u32p, _u32p]),
    "ws_debug_binning_decision": (C.c_int, [C.c_uint32, _u32p, _u32p, C.c_uint32, _u32p]),
    "ws_debug_depth_range": (C.c_int, [C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, _u32p, _u32p, _u32p]),
    "ws_debug_footprint": (C.c_int, [_u32p, C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, _u32p, _u32p]),
    "ws_sync": (C.c_int, [_P, _P]),
    "ws_context_set_host_wait": (C.c_int, [_P, C.c_int]),
    "ws_device_info": (C.c_int, [_P, C.c_char_p, C.c_size_t, _u32p, C.POINTER(C.c_uint64)]),
    "ws_device_malloc": (C.c_int, [_P, C.c_size_t, _PP]),
    "ws_device_free": (C.c_int, [_P, _P]),
    "ws_memcpy_h2d": (C.c_int, [_P, _P, _P, C.c_size_t, _P]),
    "ws_memcpy_d2h": (C.c_int, [_P, _P, _P, C.c_size_t, _P]),
    "ws_camera_fit_near_far": (C.c_int, [C.POINTER(ws_camera), C.POINTER(ws_aabb)]),
    "ws_camera_from_scene": (C.c_int, [_f32p, _f32p, C.c_float, C.c_float, C.c_uint32, C.c_uint32,
                                       C.POINTER(ws_camera)]),
    "ws_build_camera_uniform": (C.c_int, [C.POINTER(ws_camera), _u32p, C.POINTER(ws_camera_uniform)]),
    "ws_build_settings_uniform": (C.c_int, [C.POINTER(ws_splatting_args), _P, C.POINTER(ws_settings_uniform)]),
    "ws_aabb_radius": (C.c_float, [C.POINTER(ws_aabb)]),
    "ws_ply_rows_convert": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P, _P]),
    "ws_pointcloud_stats": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.POINTER(ws_aabb), C.POINTER(ws_aabb), _f32p,
                                      C.POINTER(C.c_int32), _f32p]),
    "ws_pointcloud_load_ply": (C.c_int, [_P, C.c_char_p, _PP]),
    "ws_ply_read": (C.c_int, [C.c_char_p, C.POINTER(C.POINTER(ws_ply_cloud))]),
    "ws_ply_free": (None, [C.POINTER(ws_ply_cloud)]),
    "ws_npz_read": (C.c_int, [C.c_char_p, C.POINTER(C.POINTER(ws_npz_cloud))]),
    "ws_npz_free": (None, [C.POINTER(ws_npz_cloud)]),
    "ws_pointcloud_load_npz": (C.c_int, [_P, C.c_char_p, _PP]),
    "ws_pointcloud_load": (C.c_int, [_P, C.c_char_p, _PP]),
    "ws_scene_load_json": (C.c_int, [C.c_char_p, _PP]),
    "ws_scene_from_json_text": (C.c_int, [C.c_char_p, C.c_size_t, _PP]),
    "ws_scene_destroy": (None, [_P]),
    "ws_scene_num_cameras": (C.c_uint32, [_P]),
    "ws_scene_extend": (C.c_float, [_P]),
    "ws_scene_cameras": (C.c_uint32, [_P, C.c_int, C.c_uint32, C.POINTER(ws_scene_camera)]),
    "ws_scene_get_camera": (C.c_int, [_P, C.c_uint32, C.POINTER(ws_scene_camera)]),
    "ws_scene_nearest_camera": (C.c_int, [_P, _f32p, C.c_int, _u32p]),
    "ws_download_texture_rgba8": (C.c_int, [_P, _P, C.c_int, C.c_uint32, C.c_uint32, C.c_size_t, _P, _P]),
    "ws_png_write_rgba8": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, _P, C.c_size_t]),
    "ws_render_views": (C.c_int, [_P, _P, _P, C.c_int, C.c_char_p, _u32p]),
    "ws_measure": (C.c_int, [_P, _P, _P, C.c_uint32, C.c_uint32, _f32p]),
    "ws_view_batch_create": (C.c_int, [_P, C.c_int, C.c_uint32, C.c_int, C.c_uint32, _PP]),
    "ws_view_batch_destroy": (None, [_P]),
    "ws_view_batch_frames_in_flight": (C.c_uint32, [_P]),
    "ws_view_batch_render": (C.c_int, [_P, _P, C.POINTER(ws_splatting_args), C.c_uint32, _PP, C.c_size_t, _f32p]),
    "ws_view_batch_sync": (C.c_int, [_P]),
    "ws_view_batch_errors": (C.c_int, [_P, C.POINTER(C.c_uint32), C.c_int]),
    "ws_renderer_errors": (C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_int]),
    "ws_view_batch_renderer": (C.c_void_p, [_P, C.c_uint32]),
    "ws_view_batch_host_waits": (C.c_uint32, [_P]),
    "ws_display_composite": (C.c_int, [_P, _P, C.c_int, C.c_size_t, C.c_uint32, C.c_uint32, _f32p, C.c_int, _P,
                                       C.c_size_t, _P]),
    "ws_pointcloud_create": (C.c_int, [_P, C.POINTER(ws_pointcloud_desc), _PP]),
    "ws_pointcloud_create_from_ply_rows": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.POINTER(ws_pointcloud_desc), _PP]),
    "ws_pointcloud_download": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_size_t]),
    "ws_pointcloud_destroy": (None, [_P]),
    "ws_pointcloud_num_points": (C.c_uint32, [_P]),
    "ws_pointcloud_sh_deg": (C.c_uint32, [_P]),
    "ws_pointcloud_compressed": (C.c_int, [_P]),
    "ws_pointcloud_bbox": (C.c_int, [_P, C.POINTER(ws_aabb)]),
    "ws_pointcloud_center": (C.c_int, [_P, _f32p]),
    "ws_pointcloud_up": (C.c_int, [_P, _f32p]),
    "ws_pointcloud_mip_splatting": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "ws_pointcloud_kernel_size": (C.c_int, [_P, _f32p]),
    "ws_pointcloud_background_color": (C.c_int, [_P, _f32p]),
    "ws_renderer_create": (C.c_int, [_P, C.c_int, C.c_uint32, C.c_int, _PP]),
    "ws_renderer_destroy": (None, [_P]),
    "ws_renderer_color_format": (C.c_int, [_P]),
    "ws_renderer_prepare": (C.c_int, [_P, _P, C.POINTER(ws_splatting_args), _P]),
    "ws_renderer_render": (C.c_int, [_P, _P, _f32p, _P, C.c_size_t, _P]),
    "ws_renderer_num_visible": (C.c_int, [_P, _u32p]),
    "ws_renderer_frame_stats": (C.c_int, [_P, C.POINTER(ws_frame_stats)]),
    "ws_renderer_enable_timers": (C.c_int, [_P, C.c_int]),
    "ws_renderer_stage_times": (C.c_int, [_P, C.POINTER(ws_stage_times)]),
    "ws_renderer_kernel_times": (C.c_int, [_P, C.c_uint32, C.POINTER(ws_kernel_time), _u32p]),
    "ws_renderer_download_tile_stats": (C.c_int, [_P, C.c_uint32, _P, _P, _u32p]),
    "ws_renderer_download_tile_lists": (C.c_int, [_P, C.c_uint32, _P, _P, C.c_uint32, _P, _u32p]),
    "ws_renderer_enable_capture": (C.c_int, [_P, C.c_int]),
    "ws_renderer_set_blend_mode": (C.c_int, [_P, C.c_int]),
    "ws_renderer_binning_tile": (C.c_int, [_P, _u32p, _u32p]),
    "ws_renderer_set_tile_entry_capacity": (C.c_int, [_P, C.c_uint64]),
    "ws_renderer_download_frame": (C.c_int, [_P, C.c_uint32, _P, _P, _P, _P, _u32p]),
    "ws_sorter_create": (C.c_int, [_P, C.c_uint32, _PP]),
    "ws_sorter_destroy": (None, [_P]),
    "ws_sorter_sort": (C.c_int, [_P, _P, _P, _P, C.c_uint32, _P]),
    "ws_sorter_sort_depth": (C.c_int, [_P, _P, _P, _P, _P, C.c_uint32, _P]),
    "ws_sort_selftest": (C.c_int, [_P, C.POINTER(C.c_int)]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C web-splat_amd` (or __graft_entry__.build()). "
            "websplat has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    # WEBSPLAT_LIB (an older build of the same library, for A/B measurements on one box) may predate an entry point: it is
    # then simply absent from `lib` (calling it raises AttributeError); the default library must export everything
    tolerant = bool(os.environ.get("WEBSPLAT_LIB"))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)  # AttributeError if the ABI and this stub disagree
        except AttributeError:
            if tolerant:
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(code):
    if code != WS_OK:
        raise WebSplatError(code, (lib.ws_last_error() or b"").decode("utf-8", "replace"))
    return code
