"""Python mirror of the reference's render-path API on top of the C ABI.

Names follow the reference so that tests read like web-splat code:
  WGPUContext         (src/lib.rs:57-125)        -> Context
  PerspectiveCamera   (src/camera.rs:6-35)       -> PerspectiveCamera
  Aabb                (src/pointcloud.rs:398-470)-> Aabb
  SplattingArgs       (src/renderer.rs:585-599)  -> SplattingArgs
  PointCloud          (src/pointcloud.rs:72-222) -> PointCloud
  GaussianRenderer    (src/renderer.rs:17-283)   -> GaussianRenderer
  GPURSSorter         (src/gpu_rs.rs:63-885)     -> GPURSSorter
No compute happens here; every method is one call into libwebsplat_hip.so.
"""
import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

from . import _lib as L
from ._lib import lib, check

FORMATS = {
    "rgba8unorm": (L.WS_FORMAT_RGBA8_UNORM, np.uint8, 4),
    "rgba16float": (L.WS_FORMAT_RGBA16_FLOAT, np.float16, 8),
    "rgba32float": (L.WS_FORMAT_RGBA32_FLOAT, np.float32, 16),
}


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


@dataclass
class Aabb:
    min: Sequence[float]
    max: Sequence[float]

    def to_c(self):
        a = L.ws_aabb()
        a.min[:] = [float(x) for x in self.min]
        a.max[:] = [float(x) for x in self.max]
        return a

    @staticmethod
    def from_c(a):
        return Aabb(list(a.min), list(a.max))

    def radius(self):
        a = self.to_c()
        return lib.ws_aabb_radius(C.byref(a))

    def center(self):
        return [lo + (hi - lo) / 2.0 for lo, hi in zip(self.min, self.max)]


@dataclass
class PerspectiveCamera:
    position: Sequence[float] = (0.0, 0.0, -1.0)
    rotation: Sequence[float] = (1.0, 0.0, 0.0, 0.0)  # quaternion (s, x, y, z)
    fovx: float = float(np.deg2rad(45.0))
    fovy: float = float(np.deg2rad(45.0))
    znear: float = 0.1
    zfar: float = 100.0
    fov2view_ratio: float = 1.0

    def to_c(self):
        c = L.ws_camera()
        c.position[:] = [float(x) for x in self.position]
        c.rotation[:] = [float(x) for x in self.rotation]
        c.fovx, c.fovy, c.znear, c.zfar = self.fovx, self.fovy, self.znear, self.zfar
        c.fov2view_ratio = self.fov2view_ratio
        return c

    @staticmethod
    def from_c(c):
        return PerspectiveCamera(list(c.position), list(c.rotation), c.fovx, c.fovy, c.znear, c.zfar,
                                 c.fov2view_ratio)

    def fit_near_far(self, aabb: Aabb):
        c = self.to_c()
        a = aabb.to_c()
        check(lib.ws_camera_fit_near_far(C.byref(c), C.byref(a)))
        self.znear, self.zfar = c.znear, c.zfar
        return self

    @staticmethod
    def from_scene_camera(position, rotation_rows, fx, fy, width, height):
        """scene.rs:85-108 `impl Into<PerspectiveCamera> for SceneCamera`."""
        out = L.ws_camera()
        pos = _f3(position)
        rot = (C.c_float * 9)(*[float(x) for row in rotation_rows for x in row])
        check(lib.ws_camera_from_scene(pos, rot, float(fx), float(fy), int(width), int(height), C.byref(out)))
        return PerspectiveCamera.from_c(out)

    def uniform(self, viewport):
        u = L.ws_camera_uniform()
        c = self.to_c()
        vp = (C.c_uint32 * 2)(int(viewport[0]), int(viewport[1]))
        check(lib.ws_build_camera_uniform(C.byref(c), vp, C.byref(u)))
        return u


@dataclass
class SplattingArgs:
    camera: PerspectiveCamera
    viewport: Sequence[int]
    gaussian_scaling: float = 1.0
    max_sh_deg: int = 3
    mip_splatting: Optional[bool] = None
    kernel_size: Optional[float] = None
    clipping_box: Optional[Aabb] = None
    walltime: float = 100.0  # seconds; offline callers pass Duration::from_secs(100)
    scene_center: Optional[Sequence[float]] = None
    scene_extend: Optional[float] = None
    background_color: Sequence[float] = (0.0, 0.0, 0.0, 0.0)

    def to_c(self):
        a = L.ws_splatting_args()
        a.camera = self.camera.to_c()
        a.viewport[:] = [int(self.viewport[0]), int(self.viewport[1])]
        a.gaussian_scaling = float(self.gaussian_scaling)
        a.max_sh_deg = int(self.max_sh_deg)
        a.has_mip_splatting = int(self.mip_splatting is not None)
        a.mip_splatting = int(bool(self.mip_splatting))
        a.has_kernel_size = int(self.kernel_size is not None)
        a.kernel_size = float(self.kernel_size or 0.0)
        a.has_clipping_box = int(self.clipping_box is not None)
        if self.clipping_box is not None:
            a.clipping_box = self.clipping_box.to_c()
        a.walltime_secs = float(self.walltime)
        a.has_scene_center = int(self.scene_center is not None)
        if self.scene_center is not None:
            a.scene_center[:] = [float(x) for x in self.scene_center]
        a.has_scene_extend = int(self.scene_extend is not None)
        a.scene_extend = float(self.scene_extend or 0.0)
        a.background_color[:] = [float(x) for x in self.background_color]
        return a


def stage_splat(words, viewport, tile_origin, tile_size=(16, 16)):
    """Host-side twin of the compositing pass's staging step (test hook): (rec[10], quadrant mask)."""
    w = (C.c_uint32 * 5)(*[int(x) for x in words])
    rec = (C.c_float * 10)()
    mask = C.c_uint32()
    check(lib.ws_debug_stage_splat(w, float(viewport[0]), float(viewport[1]), float(tile_origin[0]),
                                   float(tile_origin[1]), int(tile_size[0]), int(tile_size[1]), rec, C.byref(mask)))
    return np.array(rec[:], dtype=np.float32), mask.value


def footprint_tiles(words, viewport, tile_size=(32, 32)):
    """Host-side twin of the binning footprint (test hook): tile ids the splat's kept ellipse reaches, emission order."""
    w = (C.c_uint32 * 3)(*[int(x) for x in words[:3]])
    n = C.c_uint32()
    check(lib.ws_debug_footprint(w, float(viewport[0]), float(viewport[1]), int(tile_size[0]), int(tile_size[1]), 0, None,
                                 C.byref(n)))
    out = np.empty(max(n.value, 1), dtype=np.uint32)
    check(lib.ws_debug_footprint(w, float(viewport[0]), float(viewport[1]), int(tile_size[0]), int(tile_size[1]),
                                 len(out), out.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(n)))
    return out[:n.value]


def packed_rect(rect):
    """Test hook: (tiles, tiles at 2 x 2, the rectangle in units of 2 x 2 tiles) of a packed tile rectangle."""
    t, tc, rc = C.c_uint32(), C.c_uint32(), C.c_uint32()
    check(lib.ws_debug_packed_rect(int(rect), C.byref(t), C.byref(tc), C.byref(rc)))
    return t.value, tc.value, rc.value


def binning_decision(request, sums, sums_coarse):
    """Test hook: 0 / 1 = the frame bins at the compositing tile / at 2 x 2 of them, from K1's per-slot sums."""
    n = len(sums)
    a = (C.c_uint32 * max(n, 1))(*[int(x) for x in sums])
    b = (C.c_uint32 * max(n, 1))(*[int(x) for x in sums_coarse])
    sh = C.c_uint32()
    check(lib.ws_debug_binning_decision(int(request), a, b, n, C.byref(sh)))
    return sh.value


def config_from_env(env=None, **overrides):
    """The harness-side translation of the WS_* environment switches into a ws_context_config (bench.py, tests, scripts).
    The LIBRARY reads no environment variable (include/websplat.h); include/websplat_env.h is the same table for C callers.
    Keyword overrides name struct fields directly (config_from_env(blend_order=1))."""
    env = os.environ if env is None else env
    c = L.ws_context_config()
    lib.ws_context_config_init(C.byref(c))

    def num(name, field):
        v = env.get(name)
        if v not in (None, ""):
            setattr(c, field, int(v))

    for name, field in (("WS_GRAPH", "use_graph"), ("WS_DEPTH_SKIP_TOP", "depth_skip_top"), ("WS_BLEND_ORDER", "blend_order"),
                        ("WS_BLEND_SPLIT", "blend_split"), ("WS_BATCH_THREADS", "batch_threads"),
                        ("WS_BATCH_QUEUE_DEPTH", "batch_queue_depth"), ("WS_BLEND_TPW_LOG2", "blend_tpw_log2"),
                        ("WS_BLEND_LDS_PAD_KB", "blend_lds_pad_kb"), ("WS_DEBUG_CUT", "debug_cut"), ("WS_CAPTURE", "capture"),
                        ("WS_DEPTH_DIGIT_BITS", "depth_digit_bits"), ("WS_DEPTH_TILE_KPT", "depth_tile_kpt"), ("WS_BLEND_ASYNC", "blend_async"), ("WS_DSORT_FAT_GRID", "exp_dsort_fat_grid"),
                        ("WS_BLEND_VARIANT", "exp_blend_variant"), ("WS_BATCH_K1", "exp_batch_k1")):
        num(name, field)
    if env.get("WS_BLEND_DMA") not in (None, ""):
        c.exp_blend_dma = 1 if int(env["WS_BLEND_DMA"]) else 0
    bs = env.get("WS_BIN_SHIFT")
    if bs is not None:
        c.bin_request = 2 if bs == "1" else (0 if bs == "0" else 1)
    shape = env.get("WS_TILE_SHAPE")
    if shape:
        qw, _, qh = shape.partition("x")
        c.tile_qw, c.tile_qh = (int(qw), int(qh)) if qw.isdigit() and qh.isdigit() else (-1, -1)  # (the library refuses what it does not have)
    c.render_views_fast_blend = 1 if env.get("WS_RENDER_VIEWS_BLEND") == "fast" else 0
    c.ply_decode_host = 1 if env.get("WS_PLY_DECODE") == "host" else 0
    c.exp_depth_sort = {"onesweep": 1, "coop": 2}.get(env.get("WS_DEPTH_SORT", ""), 0)
    c.exp_footprint_ellipse = 1 if env.get("WS_FOOTPRINT") == "ellipse" else 0
    c.exp_tile_sort_wide = 1 if env.get("WS_TILE_SORT") == "wide" else 0
    for k, v in overrides.items():
        setattr(c, k, int(v))
    return c


class Context:
    def __init__(self, device: int = 0, config=None):
        """config: a ws_context_config (config_from_env(...)); None = this process's WS_* environment, translated HERE -- the
        harness side -- because the library itself reads no environment variable."""
        h = C.c_void_p()
        cfg = config_from_env() if config is None else config
        check(lib.ws_context_create_with_config(int(device), C.byref(cfg), C.byref(h)))
        self.handle = h
        self.device = device
        self.config = cfg

    def close(self):
        if self.handle:
            lib.ws_context_destroy(self.handle)
            self.handle = None

    def sync(self, stream=None):
        check(lib.ws_sync(self.handle, C.c_void_p(stream or 0)))

    def set_host_wait(self, mode):
        """'block' (sleep until the completion interrupt: device.poll(Wait), bin/measure.rs:147) or 'spin' (HIP's default)."""
        check(lib.ws_context_set_host_wait(self.handle, {"spin": 0, "block": 1}[mode]))

    def tile_size(self):
        """(width, height) of the binning tile in pixels (32x32 unless WS_TILE_SHAPE says otherwise)."""
        w, h = C.c_uint32(), C.c_uint32()
        check(lib.ws_context_tile_size(self.handle, C.byref(w), C.byref(h)))
        return w.value, h.value

    def device_info(self):
        name = C.create_string_buffer(128)
        cus = C.c_uint32()
        mem = C.c_uint64()
        check(lib.ws_device_info(self.handle, name, 128, C.byref(cus), C.byref(mem)))
        return {"arch": name.value.decode(), "cus": cus.value, "hbm_bytes": mem.value}

    # plain device buffers -------------------------------------------------------------------
    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        check(lib.ws_device_malloc(self.handle, int(nbytes), C.byref(p)))
        return p.value

    def free(self, ptr: int):
        check(lib.ws_device_free(self.handle, C.c_void_p(ptr)))

    def upload(self, ptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        check(lib.ws_memcpy_h2d(self.handle, C.c_void_p(ptr), arr.ctypes.data_as(C.c_void_p), arr.nbytes, None))

    def download(self, ptr: int, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        check(lib.ws_memcpy_d2h(self.handle, out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes, None))
        return out

    def sort_selftest(self) -> bool:
        ok = C.c_int(0)
        check(lib.ws_sort_selftest(self.handle, C.byref(ok)))
        return bool(ok.value)


@dataclass
class GenericGaussianPointCloud:
    """io/mod.rs:27-43: what a loader produces (host byte blobs + metadata)."""
    gaussians: np.ndarray  # uint8, N x 28 (or N x 24 compressed)
    sh_coefs: np.ndarray   # uint8, N x 96 (or packed int8)
    sh_deg: int
    num_points: int
    aabb: Aabb
    center: Sequence[float]
    compressed: bool = False
    covars: Optional[np.ndarray] = None          # uint8, M x 12
    quantization: Optional[L.ws_gaussian_quantization] = None
    up: Optional[Sequence[float]] = None
    kernel_size: Optional[float] = None
    mip_splatting: Optional[bool] = None
    background_color: Optional[Sequence[float]] = None

    @staticmethod
    def from_ply_rows(rows: np.ndarray, sh_deg: int, **meta):
        """io/ply.rs:50-100 + io/mod.rs:63-105 through the library's host-side loader code."""
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        want = 14 + 3 * (int(sh_deg) + 1) ** 2  # x,y,z, n*, f_dc*, f_rest*, opacity, scale*, rot*  (io/ply.rs:54-88)
        if rows.ndim != 2 or rows.shape[1] != want:
            raise ValueError(f"from_ply_rows: sh_deg {sh_deg} needs rows of {want} floats, got shape {rows.shape}")
        n = rows.shape[0]
        g = np.empty((n, 28), dtype=np.uint8)
        s = np.empty((n, 96), dtype=np.uint8)
        check(lib.ws_ply_rows_convert(rows.ctypes.data_as(C.c_void_p), n, int(sh_deg),
                                      g.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p)))
        aabb, center, up = pointcloud_stats(g, 28, Aabb([0, 0, 0], [0, 0, 0]))
        return GenericGaussianPointCloud(g, s, sh_deg, n, aabb, center, up=up, **meta)


def read_ply(path: str) -> GenericGaussianPointCloud:
    """io/ply.rs:28-196 PlyReader::read + io/mod.rs:63-105 through the library's host-side reader (no GPU)."""
    pp = C.POINTER(L.ws_ply_cloud)()
    check(lib.ws_ply_read(str(path).encode(), C.byref(pp)))
    try:
        c = pp.contents
        if c.num_points == 0:  # an empty vertex list reads fine in the reference too (it cannot be uploaded)
            g, sh = np.empty((0, 28), np.uint8), np.empty((0, 96), np.uint8)
        else:
            g = np.ctypeslib.as_array((C.c_uint8 * c.gaussians_bytes).from_address(c.gaussians)).copy().reshape(-1, 28)
            sh = np.ctypeslib.as_array((C.c_uint8 * c.sh_coefs_bytes).from_address(c.sh_coefs)).copy().reshape(-1, 96)
        return GenericGaussianPointCloud(
            g, sh, int(c.sh_deg), int(c.num_points), Aabb(list(c.bbox.min), list(c.bbox.max)), list(c.center),
            up=list(c.up) if c.has_up else None,
            kernel_size=c.kernel_size if c.has_kernel_size else None,
            mip_splatting=bool(c.mip_splatting) if c.has_mip_splatting else None,
            background_color=list(c.background_color) if c.has_background_color else None)
    finally:
        lib.ws_ply_free(pp)


def read_npz(path: str) -> GenericGaussianPointCloud:
    """io/npz.rs:59-225 NpzReader::read + io/mod.rs:107-150 new_compressed, through the library's native reader."""
    pp = C.POINTER(L.ws_npz_cloud)()
    check(lib.ws_npz_read(str(path).encode(), C.byref(pp)))
    try:
        c = pp.contents
        g = np.ctypeslib.as_array((C.c_uint8 * c.gaussians_bytes).from_address(c.gaussians)).copy().reshape(-1, 24)
        sh = np.ctypeslib.as_array((C.c_uint8 * c.sh_coefs_bytes).from_address(c.sh_coefs)).copy()
        cv = np.ctypeslib.as_array((C.c_uint8 * c.covars_bytes).from_address(c.covars)).copy().reshape(-1, 12)
        q = L.ws_gaussian_quantization()
        C.memmove(C.byref(q), C.byref(c.quantization), C.sizeof(q))
        aabb, center, up = pointcloud_stats(g, 24, Aabb([-1, -1, -1], [1, 1, 1]))  # Aabb::unit(), io/mod.rs:119
        return GenericGaussianPointCloud(
            g, sh, int(c.sh_deg), int(c.num_points), aabb, center, compressed=True, covars=cv, quantization=q, up=up,
            kernel_size=c.kernel_size if c.has_kernel_size else None,
            mip_splatting=bool(c.mip_splatting) if c.has_mip_splatting else None,
            background_color=list(c.background_color) if c.has_background_color else None)
    finally:
        lib.ws_npz_free(pp)


def write_png(path: str, rgba: np.ndarray):
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    h, w = rgba.shape[:2]
    check(lib.ws_png_write_rgba8(str(path).encode(), w, h, rgba.ctypes.data_as(C.c_void_p), w * 4))


@dataclass
class SceneCamera:
    """scene.rs:13-24."""
    id: int
    img_name: str
    width: int
    height: int
    position: Sequence[float]
    rotation: Sequence[Sequence[float]]
    fx: float
    fy: float
    split: str = "train"

    @staticmethod
    def from_c(c):
        rot = [list(c.rotation[3 * k:3 * k + 3]) for k in range(3)]
        return SceneCamera(c.id, c.img_name.decode(), c.width, c.height, list(c.position), rot, c.fx, c.fy,
                           "test" if c.split == L.WS_SPLIT_TEST else "train")

    def to_perspective(self) -> "PerspectiveCamera":
        return PerspectiveCamera.from_scene_camera(self.position, self.rotation, self.fx, self.fy, self.width, self.height)


_SPLITS = {None: L.WS_SPLIT_ALL, "train": L.WS_SPLIT_TRAIN, "test": L.WS_SPLIT_TEST}


class Scene:
    """scene.rs:113-194 Scene."""

    def __init__(self, handle):
        self.handle = handle

    @staticmethod
    def from_json(path: str):
        h = C.c_void_p()
        check(lib.ws_scene_load_json(str(path).encode(), C.byref(h)))
        return Scene(h)

    @staticmethod
    def from_json_text(text: str):
        raw = text.encode()
        h = C.c_void_p()
        check(lib.ws_scene_from_json_text(raw, len(raw), C.byref(h)))
        return Scene(h)

    def close(self):
        if self.handle:
            lib.ws_scene_destroy(self.handle)
            self.handle = None

    def num_cameras(self):
        return lib.ws_scene_num_cameras(self.handle)

    def extend(self):
        return lib.ws_scene_extend(self.handle)

    def cameras(self, split=None):
        n = lib.ws_scene_cameras(self.handle, _SPLITS[split], 0, None)
        buf = (L.ws_scene_camera * max(n, 1))()
        lib.ws_scene_cameras(self.handle, _SPLITS[split], n, buf)
        return [SceneCamera.from_c(buf[i]) for i in range(n)]

    def camera(self, cam_id: int):
        c = L.ws_scene_camera()
        return SceneCamera.from_c(c) if lib.ws_scene_get_camera(self.handle, int(cam_id), C.byref(c)) else None

    def nearest_camera(self, pos, split=None):
        out = C.c_uint32()
        return out.value if lib.ws_scene_nearest_camera(self.handle, _f3(pos), _SPLITS[split], C.byref(out)) else None


def render_views(ctx: "Context", pc: "PointCloud", scene: Scene, split: str, out_dir: str) -> int:
    """bin/render.rs:33-128 render_views."""
    n = C.c_uint32()
    check(lib.ws_render_views(ctx.handle, pc.handle, scene.handle, _SPLITS[split], str(out_dir).encode(), C.byref(n)))
    return n.value


def measure(ctx: "Context", pc: "PointCloud", scene: Scene, num_samples: int = 10, frames_in_flight: int = 1) -> float:
    """bin/measure.rs:27-154: average FPS over the training cameras at 2048x2048."""
    fps = C.c_float()
    check(lib.ws_measure(ctx.handle, pc.handle, scene.handle, int(num_samples), int(frames_in_flight), C.byref(fps)))
    return fps.value


def pointcloud_stats(gaussians: np.ndarray, stride: int, start: Aabb):
    start_c = start.to_c()
    bbox = L.ws_aabb()
    center = (C.c_float * 3)()
    has_up = C.c_int32()
    up = (C.c_float * 3)()
    n = gaussians.shape[0]
    check(lib.ws_pointcloud_stats(gaussians.ctypes.data_as(C.c_void_p), n, stride, C.byref(start_c), C.byref(bbox),
                                  center, C.byref(has_up), up))
    return Aabb.from_c(bbox), list(center), (list(up) if has_up.value else None)


class PointCloud:
    """pointcloud.rs:99-222 PointCloud::new(device, GenericGaussianPointCloud)."""

    def __init__(self, ctx: Context, pc: GenericGaussianPointCloud = None, _handle=None):
        self.ctx = ctx
        if _handle is not None:
            self.handle = _handle
            return
        d = L.ws_pointcloud_desc()
        g = np.ascontiguousarray(pc.gaussians)
        s = np.ascontiguousarray(pc.sh_coefs)
        d.num_points = pc.num_points
        d.sh_deg = pc.sh_deg
        d.compressed = int(pc.compressed)
        d.gaussians = g.ctypes.data
        d.gaussians_bytes = g.nbytes
        d.sh_coefs = s.ctypes.data
        d.sh_coefs_bytes = s.nbytes
        keep = [g, s]
        if pc.compressed:
            cv = np.ascontiguousarray(pc.covars)
            keep.append(cv)
            d.covars = cv.ctypes.data
            d.covars_bytes = cv.nbytes
            d.quantization = C.pointer(pc.quantization)
        d.bbox = pc.aabb.to_c()
        d.center[:] = [float(x) for x in pc.center]
        d.has_up = int(pc.up is not None)
        if pc.up is not None:
            d.up[:] = [float(x) for x in pc.up]
        d.has_mip_splatting = int(pc.mip_splatting is not None)
        d.mip_splatting = int(bool(pc.mip_splatting))
        d.has_kernel_size = int(pc.kernel_size is not None)
        d.kernel_size = float(pc.kernel_size or 0.0)
        d.has_background_color = int(pc.background_color is not None)
        if pc.background_color is not None:
            d.background_color[:] = [float(x) for x in pc.background_color]
        h = C.c_void_p()
        check(lib.ws_pointcloud_create(ctx.handle, C.byref(d), C.byref(h)))
        self.handle = h

    @staticmethod
    def from_ply_rows(ctx: Context, rows: np.ndarray, sh_deg: int, kernel_size=None, mip_splatting=None,
                      background_color=None):
        """Raw PLY vertex rows -> resident scene with the conversion (io/ply.rs:50-100) on the GPU."""
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        want = 14 + 3 * (int(sh_deg) + 1) ** 2
        if rows.ndim != 2 or rows.shape[1] != want:
            raise ValueError(f"from_ply_rows: sh_deg {sh_deg} needs rows of {want} floats, got shape {rows.shape}")
        d = L.ws_pointcloud_desc()
        d.has_mip_splatting = int(mip_splatting is not None)
        d.mip_splatting = int(bool(mip_splatting))
        d.has_kernel_size = int(kernel_size is not None)
        d.kernel_size = float(kernel_size or 0.0)
        d.has_background_color = int(background_color is not None)
        if background_color is not None:
            d.background_color[:] = [float(x) for x in background_color]
        h = C.c_void_p()
        check(lib.ws_pointcloud_create_from_ply_rows(ctx.handle, rows.ctypes.data_as(C.c_void_p), rows.shape[0], int(sh_deg),
                                                     C.byref(d), C.byref(h)))
        return PointCloud(ctx, _handle=h)

    def download(self):
        """(gaussians N x 28 | N x 24, sh N x 96 | None): the resident scene as loader blobs (parity tooling)."""
        n = self.num_points()
        if self.compressed():
            g = np.empty((n, 24), dtype=np.uint8)
            s = np.empty((1,), dtype=np.uint8)
            check(lib.ws_pointcloud_download(self.handle, g.ctypes.data_as(C.c_void_p), g.nbytes, s.ctypes.data_as(C.c_void_p), 0))
            return g, None
        g = np.empty((n, 28), dtype=np.uint8)
        s = np.empty((n, 96), dtype=np.uint8)
        check(lib.ws_pointcloud_download(self.handle, g.ctypes.data_as(C.c_void_p), g.nbytes, s.ctypes.data_as(C.c_void_p), s.nbytes))
        return g, s

    @staticmethod
    def load_ply(ctx: Context, path: str):
        h = C.c_void_p()
        check(lib.ws_pointcloud_load_ply(ctx.handle, path.encode(), C.byref(h)))
        return PointCloud(ctx, _handle=h)

    @staticmethod
    def load_npz(ctx: Context, path: str):
        h = C.c_void_p()
        check(lib.ws_pointcloud_load_npz(ctx.handle, str(path).encode(), C.byref(h)))
        return PointCloud(ctx, _handle=h)

    @staticmethod
    def load(ctx: Context, path: str):
        """io/mod.rs:45-61 GenericGaussianPointCloud::load (magic-byte sniffing) + PointCloud::new."""
        h = C.c_void_p()
        check(lib.ws_pointcloud_load(ctx.handle, str(path).encode(), C.byref(h)))
        return PointCloud(ctx, _handle=h)

    def close(self):
        if self.handle:
            lib.ws_pointcloud_destroy(self.handle)
            self.handle = None

    def num_points(self):
        return lib.ws_pointcloud_num_points(self.handle)

    def sh_deg(self):
        return lib.ws_pointcloud_sh_deg(self.handle)

    def compressed(self):
        return bool(lib.ws_pointcloud_compressed(self.handle))

    def bbox(self) -> Aabb:
        a = L.ws_aabb()
        check(lib.ws_pointcloud_bbox(self.handle, C.byref(a)))
        return Aabb.from_c(a)

    def center(self):
        c = (C.c_float * 3)()
        check(lib.ws_pointcloud_center(self.handle, c))
        return list(c)

    def up(self):
        c = (C.c_float * 3)()
        return list(c) if lib.ws_pointcloud_up(self.handle, c) else None

    def mip_splatting(self):
        v = C.c_int32()
        return bool(v.value) if lib.ws_pointcloud_mip_splatting(self.handle, C.byref(v)) else None

    def dilation_kernel_size(self):
        v = C.c_float()
        return v.value if lib.ws_pointcloud_kernel_size(self.handle, C.byref(v)) else None

    def background_color(self):
        c = (C.c_float * 3)()
        return list(c) if lib.ws_pointcloud_background_color(self.handle, c) else None

    def settings_uniform(self, args: SplattingArgs):
        u = L.ws_settings_uniform()
        a = args.to_c()
        check(lib.ws_build_settings_uniform(C.byref(a), self.handle, C.byref(u)))
        return u


class GaussianRenderer:
    """renderer.rs:33-283.  `prepare` + `render` enqueue on a HIP stream; nothing syncs except the
    read-back helpers (num_visible_points, frame_stats, stage_times, download_*)."""

    def __init__(self, ctx: Context, color_format: str = "rgba32float", sh_deg: int = 3, compressed: bool = False):
        self.ctx = ctx
        self.color_format_name = color_format
        fmt, self.np_dtype, self.texel_bytes = FORMATS[color_format]
        h = C.c_void_p()
        check(lib.ws_renderer_create(ctx.handle, fmt, int(sh_deg), int(compressed), C.byref(h)))
        self.handle = h
        self._own_target = None
        self._own_target_shape = None

    def close(self):
        if self._own_target:
            self.ctx.free(self._own_target)
            self._own_target = None
        if self.handle:
            lib.ws_renderer_destroy(self.handle)
            self.handle = None

    def color_format(self):
        return self.color_format_name

    def enable_timers(self, on=True):
        """0/False = off, 1/True = stage times, 2 = additionally per-kernel times."""
        check(lib.ws_renderer_enable_timers(self.handle, int(on)))

    def kernel_times(self):
        """[(label, ms)] of every kernel launch of the last frame, launch order (needs enable_timers(2))."""
        n = C.c_uint32()
        buf = (L.ws_kernel_time * 64)()
        check(lib.ws_renderer_kernel_times(self.handle, 64, buf, C.byref(n)))
        return [(buf[i].name.decode(), buf[i].ms) for i in range(min(n.value, 64))]

    def tile_lists(self):
        """(begin[T], end[T], entries[D]): tile t's far -> near splat list is entries[begin[t]:end[t]] (store indices)."""
        nt = C.c_uint32()
        check(lib.ws_renderer_download_tile_stats(self.handle, 0, None, None, C.byref(nt)))
        d = C.c_uint32()
        check(lib.ws_renderer_download_tile_lists(self.handle, 0, None, None, 0, None, C.byref(d)))
        begin = np.empty(nt.value, dtype=np.uint32)
        end = np.empty(nt.value, dtype=np.uint32)
        entries = np.empty(max(d.value, 1), dtype=np.uint32)
        check(lib.ws_renderer_download_tile_lists(self.handle, nt.value, begin.ctypes.data_as(C.c_void_p),
                                                  end.ctypes.data_as(C.c_void_p), entries.size,
                                                  entries.ctypes.data_as(C.c_void_p), C.byref(d)))
        return begin, end, entries[:d.value]

    def wave_stats(self):
        """[tiles, 17] uint32 (capture mode): records composited per wave; column 16 = lock-step cost (see websplat.h)."""
        nt = C.c_uint32()
        check(lib.ws_renderer_download_tile_stats(self.handle, 0, None, None, C.byref(nt)))
        out = np.zeros((nt.value, 17), dtype=np.uint32)
        check(lib.ws_renderer_download_wave_stats(self.handle, nt.value, out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    def depth_sort_passes(self) -> int:
        """digit passes the last prepared frame's depth sort executed (websplat.h)."""
        v = C.c_uint32()
        check(lib.ws_renderer_depth_sort_passes(self.handle, C.byref(v)))
        return v.value

    def enable_frame_trace(self, frames: int):
        """K1 and the blend of the next `frames` frames leave (start, end) on the device clock (websplat.h)."""
        check(lib.ws_renderer_enable_frame_trace(self.handle, int(frames)))

    def frame_trace(self):
        """[frames, 4] uint64 ticks of the 100-MHz device clock: K1 start, K1 end, blend start, blend end (syncs)."""
        n = C.c_uint32()
        check(lib.ws_renderer_download_frame_trace(self.handle, 0, None, C.byref(n)))
        out = np.zeros((n.value, 4), dtype=np.uint64)
        if n.value:
            check(lib.ws_renderer_download_frame_trace(self.handle, n.value, out.ctypes.data_as(C.c_void_p), C.byref(n)))
        return out

    def depth_sort_digit_bits(self) -> int:
        """digit width (8 | 9) of the last prepared frame's depth sort (websplat.h)."""
        v = C.c_uint32()
        check(lib.ws_renderer_depth_sort_digit_bits(self.handle, C.byref(v)))
        return v.value

    def blend_order(self):
        """[blocks, 4] uint32 (tx | ty << 16, begin, end, 0): the compositing workgroups' tiles, longest list first; empty when
        the last prepared frame was not ordered (websplat.h)."""
        nb = C.c_uint32()
        check(lib.ws_renderer_download_blend_order(self.handle, 0, None, C.byref(nb)))
        out = np.zeros((nb.value, 4), dtype=np.uint32)
        if nb.value:
            check(lib.ws_renderer_download_blend_order(self.handle, nb.value, out.ctypes.data_as(C.c_void_p), C.byref(nb)))
        return out

    def enable_blend_timing(self, on=True):
        check(lib.ws_renderer_enable_blend_timing(self.handle, 1 if on else 0))

    def blend_timing(self):
        """[tiles, 16 waves, 16 words] uint32 of the last render() under enable_blend_timing (layout: websplat.h)."""
        nt = C.c_uint32()
        check(lib.ws_renderer_download_blend_timing(self.handle, 0, None, C.byref(nt)))
        out = np.zeros((nt.value, 16, 16), dtype=np.uint32)
        check(lib.ws_renderer_download_blend_timing(self.handle, nt.value, out.ctypes.data_as(C.c_void_p), None))
        return out

    def tile_stats(self, with_consumed=False):
        nt = C.c_uint32()
        check(lib.ws_renderer_download_tile_stats(self.handle, 0, None, None, C.byref(nt)))
        ll = np.empty(nt.value, dtype=np.uint32)
        cons = np.empty(nt.value, dtype=np.uint32) if with_consumed else None
        check(lib.ws_renderer_download_tile_stats(
            self.handle, nt.value, ll.ctypes.data_as(C.c_void_p),
            cons.ctypes.data_as(C.c_void_p) if cons is not None else None, C.byref(nt)))
        return {"list_len": ll, "consumed": cons}

    def enable_capture(self, on=True):
        check(lib.ws_renderer_enable_capture(self.handle, int(on)))

    def binning_tile(self):
        """(width, height) in pixels of the binning tile the last prepared frame used (the device decides per frame)."""
        w, h = C.c_uint32(), C.c_uint32()
        check(lib.ws_renderer_binning_tile(self.handle, C.byref(w), C.byref(h)))
        return w.value, h.value

    def set_blend_mode(self, mode="fast"):
        """"fast" (front to back, early-out, one rounding at the store) or "target" (the reference's fixed-function blend
        literally: back to front, the destination rounded to the target's precision after every splat)."""
        check(lib.ws_renderer_set_blend_mode(self.handle, {"fast": 0, "target": 1, "fast_exact_cut": 2}[mode]))

    def set_tile_entry_capacity(self, entries: int):
        check(lib.ws_renderer_set_tile_entry_capacity(self.handle, int(entries)))

    def prepare(self, pc: PointCloud, args: SplattingArgs, stream=None):
        a = args.to_c()
        check(lib.ws_renderer_prepare(self.handle, pc.handle, C.byref(a), C.c_void_p(stream or 0)))
        self._viewport = (int(args.viewport[0]), int(args.viewport[1]))

    def render(self, pc: PointCloud, target_ptr: int = None, pitch: int = None, background=(0.0, 0.0, 0.0, 0.0),
               stream=None):
        w, h = self._viewport
        if target_ptr is None:
            if self._own_target_shape != (w, h):
                if self._own_target:
                    self.ctx.free(self._own_target)
                self._own_target = self.ctx.malloc(w * h * self.texel_bytes)
                self._own_target_shape = (w, h)
            target_ptr = self._own_target
        if pitch is None:
            pitch = w * self.texel_bytes
        bg = (C.c_float * 4)(*[float(x) for x in background])
        check(lib.ws_renderer_render(self.handle, pc.handle, bg, C.c_void_p(target_ptr), pitch, C.c_void_p(stream or 0)))
        return target_ptr

    def download_target_rgba8(self) -> np.ndarray:
        """bin/render.rs:187-246 download_texture of the renderer-owned target: H x W x 4 uint8 (truncating)."""
        w, h = self._own_target_shape
        out = np.empty((h, w, 4), dtype=np.uint8)
        check(lib.ws_download_texture_rgba8(self.ctx.handle, C.c_void_p(self._own_target), FORMATS[self.color_format_name][0],
                                            w, h, w * self.texel_bytes, out.ctypes.data_as(C.c_void_p), None))
        return out

    def display(self, background=(0.0, 0.0, 0.0, 1.0), surface="rgba8unorm") -> np.ndarray:
        """Display::render (renderer.rs:548-582) of the renderer-owned target into an 8-bit surface: H x W x 4."""
        w, h = self._own_target_shape
        dst = self.ctx.malloc(w * h * 4)
        try:
            bg = (C.c_float * 4)(*[float(x) for x in background])
            sf = {"rgba8unorm": L.WS_SURFACE_RGBA8_UNORM, "bgra8unorm": L.WS_SURFACE_BGRA8_UNORM}[surface]
            check(lib.ws_display_composite(self.ctx.handle, C.c_void_p(self._own_target), FORMATS[self.color_format_name][0],
                                           w * self.texel_bytes, w, h, bg, sf, C.c_void_p(dst), w * 4, None))
            self.ctx.sync()
            return self.ctx.download(dst, (h, w, 4), np.uint8)
        finally:
            self.ctx.free(dst)

    def download_target(self) -> np.ndarray:
        """download_texture (bin/render.rs:187-246) for the renderer-owned target: H x W x 4."""
        w, h = self._own_target_shape
        self.ctx.sync()
        return self.ctx.download(self._own_target, (h, w, 4), self.np_dtype)

    def num_visible_points(self) -> int:
        v = C.c_uint32()
        check(lib.ws_renderer_num_visible(self.handle, C.byref(v)))
        return v.value

    def frame_stats(self):
        s = L.ws_frame_stats()
        check(lib.ws_renderer_frame_stats(self.handle, C.byref(s)))
        return {"num_visible": s.num_visible, "num_tile_entries": s.num_tile_entries,
                "tile_entries_capacity": s.tile_entries_capacity, "overflow": s.overflow}

    def errors(self, reset=False):
        """(bits, entries_needed): error bits of every frame drawn since creation / the last reset (syncs)."""
        bits, need = C.c_uint32(), C.c_uint32()
        check(lib.ws_renderer_errors(self.handle, C.byref(bits), C.byref(need), int(bool(reset))))
        return bits.value, need.value

    def stage_times(self):
        s = L.ws_stage_times()
        check(lib.ws_renderer_stage_times(self.handle, C.byref(s)))
        return {"preprocess": s.preprocess_ms, "sorting": s.sorting_ms, "binning": s.binning_ms,
                "rasterization": s.rasterization_ms}

    def download_frame(self, with_src_index=False):
        v = self.num_visible_points()
        splats = np.empty((v, 20), dtype=np.uint8)
        keys = np.empty(v, dtype=np.uint32)
        sorted_idx = np.empty(v, dtype=np.uint32)
        src = np.empty(v, dtype=np.uint32) if with_src_index else None
        nv = C.c_uint32()
        check(lib.ws_renderer_download_frame(
            self.handle, v, splats.ctypes.data_as(C.c_void_p), keys.ctypes.data_as(C.c_void_p),
            src.ctypes.data_as(C.c_void_p) if src is not None else None, sorted_idx.ctypes.data_as(C.c_void_p),
            C.byref(nv)))
        return {"num_visible": nv.value, "splats": splats, "keys": keys, "sorted": sorted_idx, "src_index": src}


class ViewBatch:
    """Several frames in flight over one resident scene (ws_view_batch_*): the unit a rank renders its shard of views with."""

    def __init__(self, ctx: Context, color_format: str = "rgba32float", sh_deg: int = 3, compressed: bool = False,
                 frames_in_flight: int = 4):
        self.ctx = ctx
        self.color_format_name = color_format
        fmt, self.np_dtype, self.texel_bytes = FORMATS[color_format]
        h = C.c_void_p()
        check(lib.ws_view_batch_create(ctx.handle, fmt, int(sh_deg), int(compressed), int(frames_in_flight), C.byref(h)))
        self.handle = h
        self.frames_in_flight = lib.ws_view_batch_frames_in_flight(h)

    def close(self):
        if self.handle:
            lib.ws_view_batch_destroy(self.handle)
            self.handle = None

    @staticmethod
    def pack_views(views):
        """SplattingArgs list -> contiguous ws_splatting_args array (convert once, render many times)."""
        arr = (L.ws_splatting_args * len(views))()
        for i, v in enumerate(views):
            arr[i] = v.to_c()
        return arr

    def render(self, pc: "PointCloud", views, target_ptrs, row_pitch: int, background=(0.0, 0.0, 0.0, 0.0)):
        """Enqueue len(views) frames (views: list of SplattingArgs or a pack_views() array); returns immediately."""
        arr = views if isinstance(views, C.Array) else ViewBatch.pack_views(views)
        n = len(arr)
        tp = target_ptrs if isinstance(target_ptrs, C.Array) else (C.c_void_p * n)(*[int(p) for p in target_ptrs])
        bg = (C.c_float * 4)(*[float(x) for x in background])
        check(lib.ws_view_batch_render(self.handle, pc.handle, arr, n, tp, int(row_pitch), bg))

    def sync(self):
        check(lib.ws_view_batch_sync(self.handle))

    def errors(self, reset=False):
        """OR of the slots' sticky error bits (tile-entry overflow, look-back time-outs); syncs."""
        bits = C.c_uint32()
        check(lib.ws_view_batch_errors(self.handle, C.byref(bits), int(bool(reset))))
        return bits.value

    def host_waits(self) -> int:
        """times render() slept because a slot's host side was queue_depth frames ahead of the device (websplat.h)."""
        return int(lib.ws_view_batch_host_waits(self.handle))

    def renderer(self, slot: int) -> "GaussianRenderer":
        """A non-owning view of slot's renderer (frame_stats, timers)."""
        r = GaussianRenderer.__new__(GaussianRenderer)
        r.ctx = self.ctx
        r.color_format_name = self.color_format_name
        _, r.np_dtype, r.texel_bytes = FORMATS[self.color_format_name]
        r.handle = C.c_void_p(lib.ws_view_batch_renderer(self.handle, int(slot)))
        r._own_target = None
        r._own_target_shape = None
        r.close = lambda: None
        return r


class GPURSSorter:
    """gpu_rs.rs: GPURSSorter::new + create_sort_stuff(max_n); sort() = record_sort / record_sort_indirect."""

    def __init__(self, ctx: Context, max_n: int):
        self.ctx = ctx
        h = C.c_void_p()
        check(lib.ws_sorter_create(ctx.handle, int(max_n), C.byref(h)))
        self.handle = h

    def close(self):
        if self.handle:
            lib.ws_sorter_destroy(self.handle)
            self.handle = None

    def sort(self, d_keys: int, d_payload: int, n: int, d_count: int = None, stream=None):
        check(lib.ws_sorter_sort(self.handle, C.c_void_p(d_keys), C.c_void_p(d_payload),
                                 C.c_void_p(d_count) if d_count else None, int(n), C.c_void_p(stream or 0)))

    def sort_depth(self, d_keys: int, d_payload: int, n: int, d_aux: int = None, d_count: int = None, stream=None):
        """The renderer's depth sort as a stand-alone call: same result as sort(), a 4-byte companion value rides along."""
        check(lib.ws_sorter_sort_depth(self.handle, C.c_void_p(d_keys), C.c_void_p(d_payload),
                                       C.c_void_p(d_aux) if d_aux else None, C.c_void_p(d_count) if d_count else None,
                                       int(n), C.c_void_p(stream or 0)))

    def sort_host(self, keys: np.ndarray, payload: np.ndarray, count: int = None, depth: bool = False, aux: np.ndarray = None):
        """Convenience for tests: upload, sort, download. `count` exercises the device-side count path; `depth`
        selects the depth-sort specialisation, `aux` its companion values (returned as a third array)."""
        n = keys.shape[0]
        dk = self.ctx.malloc(max(n, 1) * 4)
        dv = self.ctx.malloc(max(n, 1) * 4)
        da = self.ctx.malloc(max(n, 1) * 4) if aux is not None else None
        dc = None
        try:
            self.ctx.upload(dk, keys.astype(np.uint32))
            self.ctx.upload(dv, payload.astype(np.uint32))
            if aux is not None:
                self.ctx.upload(da, aux.astype(np.uint32))
            if count is not None:
                dc = self.ctx.malloc(4)
                self.ctx.upload(dc, np.array([count], dtype=np.uint32))
            if depth:
                self.sort_depth(dk, dv, n, da, dc)
            else:
                self.sort(dk, dv, n, dc)
            self.ctx.sync()
            out = (self.ctx.download(dk, (n,), np.uint32), self.ctx.download(dv, (n,), np.uint32))
            if aux is not None:
                out = out + (self.ctx.download(da, (n,), np.uint32),)
            return out
        finally:
            self.ctx.free(dk)
            self.ctx.free(dv)
            if da:
                self.ctx.free(da)
            if dc:
                self.ctx.free(dc)
