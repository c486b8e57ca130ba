"""ctypes wrapper of oracle/libws_oracle.so -- TEST INFRASTRUCTURE ONLY (see oracle/ws_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libws_oracle.so")


def build(force=False):
    src = os.path.join(ORACLE_DIR, "ws_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])


BUILD_FLAGS = "-O2 -ffp-contract=off -fno-fast-math"


def _build_native():
    """bench.py's cpu_baseline leg (WS_ORACLE_NATIVE=1): a -O3 -march=native build of the same source, compiled ON
    THE HOST THAT RUNS IT into a temporary directory -- a -march=native object must never travel between machines.
    Still -ffp-contract=off / no fast-math: vectorisation does not change IEEE results, so it stays the same oracle."""
    import tempfile
    out = os.path.join(tempfile.mkdtemp(prefix="ws_oracle_native_"), "libws_oracle_native.so")
    flags = ["-O3", "-march=native", "-std=c11", "-fPIC", "-fopenmp", "-ffp-contract=off", "-fno-fast-math"]
    subprocess.check_call(["gcc", *flags, "-shared", "-o", out, os.path.join(ORACLE_DIR, "ws_oracle.c"), "-lm"],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out, "-O3 -march=native -ffp-contract=off -fno-fast-math (built on this host)"


build()
if os.environ.get("WS_ORACLE_NATIVE") == "1":
    try:
        LIB_PATH, BUILD_FLAGS = _build_native()
    except Exception:  # noqa: BLE001  (no compiler on the host: keep the portable build)
        pass
_lib = C.CDLL(LIB_PATH)


class Camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("rotation", C.c_float * 4), ("fovx", C.c_float), ("fovy", C.c_float),
                ("znear", C.c_float), ("zfar", C.c_float), ("fov2view_ratio", C.c_float)]


class Aabb(C.Structure):
    _fields_ = [("min", C.c_float * 3), ("max", C.c_float * 3)]


class CameraUniform(C.Structure):
    _fields_ = [("view", C.c_float * 16), ("view_inv", C.c_float * 16), ("proj", C.c_float * 16),
                ("proj_inv", C.c_float * 16), ("viewport", C.c_float * 2), ("focal", C.c_float * 2)]


class SettingsUniform(C.Structure):
    _fields_ = [("clip_min", C.c_float * 4), ("clip_max", C.c_float * 4), ("gaussian_scaling", C.c_float),
                ("max_sh_deg", C.c_uint32), ("mip_splatting", C.c_uint32), ("kernel_size", C.c_float),
                ("walltime", C.c_float), ("scene_extend", C.c_float), ("_pad", C.c_uint32 * 2),
                ("scene_center", C.c_float * 4)]


class Quantization(C.Structure):
    _fields_ = [("zero_point", C.c_int32), ("scale", C.c_float), ("_pad", C.c_uint32 * 2)]


class GaussianQuantization(C.Structure):
    _fields_ = [("color_dc", Quantization), ("color_rest", Quantization), ("opacity", Quantization),
                ("scaling_factor", Quantization)]


_vp = C.c_void_p
_lib.wso_f32_to_f16.restype = C.c_uint16
_lib.wso_f32_to_f16.argtypes = [C.c_float]
_lib.wso_f16_to_f32.restype = C.c_float
_lib.wso_f16_to_f32.argtypes = [C.c_uint16]
_lib.wso_camera_uniform_build.argtypes = [C.POINTER(Camera), C.c_uint32, C.c_uint32, C.POINTER(CameraUniform)]
_lib.wso_fit_near_far.argtypes = [C.POINTER(Camera), C.POINTER(Aabb)]
_lib.wso_aabb_radius.restype = C.c_float
_lib.wso_aabb_radius.argtypes = [C.POINTER(Aabb)]
_lib.wso_scene_camera_to_perspective.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float,
                                                 C.c_uint32, C.c_uint32, C.POINTER(Camera)]
_lib.wso_ply_rows_convert.argtypes = [_vp, C.c_uint32, C.c_uint32, _vp, _vp]
_lib.wso_pointcloud_stats.restype = C.c_int
_lib.wso_pointcloud_stats.argtypes = [_vp, C.c_uint32, C.c_uint32, C.POINTER(Aabb), C.POINTER(Aabb),
                                      C.POINTER(C.c_float), C.POINTER(C.c_float)]
_lib.wso_preprocess.restype = C.c_uint32
_lib.wso_preprocess.argtypes = [_vp, _vp, C.c_uint32, C.POINTER(CameraUniform), C.POINTER(SettingsUniform), _vp, _vp,
                                _vp]
_lib.wso_preprocess_compressed.restype = C.c_uint32
_lib.wso_preprocess_compressed.argtypes = [_vp, _vp, _vp, C.POINTER(GaussianQuantization), C.c_uint32, C.c_uint32,
                                           C.POINTER(CameraUniform), C.POINTER(SettingsUniform), _vp, _vp, _vp]
_lib.wso_sort_pairs.argtypes = [_vp, _vp, C.c_uint32]
_lib.wso_render.argtypes = [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float), C.c_int, _vp]
_lib.wso_num_threads.restype = C.c_int
_lib.wso_sigmoid.restype = C.c_float
_lib.wso_sigmoid.argtypes = [C.c_float]
_lib.wso_build_cov.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def f32_to_f16(x):
    return _lib.wso_f32_to_f16(float(x))


def f16_to_f32(h):
    return _lib.wso_f16_to_f32(int(h))


def num_threads():
    return _lib.wso_num_threads()


def copy_struct(dst_cls, src):
    """Byte-copy a ctypes struct of identical layout (e.g. the library's uniform into the oracle's)."""
    assert C.sizeof(dst_cls) == C.sizeof(src)
    dst = dst_cls()
    C.memmove(C.byref(dst), C.byref(src), C.sizeof(src))
    return dst


def make_camera(position, rotation, fovx, fovy, znear, zfar, ratio=1.0):
    c = Camera()
    c.position[:] = [float(x) for x in position]
    c.rotation[:] = [float(x) for x in rotation]
    c.fovx, c.fovy, c.znear, c.zfar, c.fov2view_ratio = fovx, fovy, znear, zfar, ratio
    return c


def make_aabb(lo, hi):
    a = Aabb()
    a.min[:] = [float(x) for x in lo]
    a.max[:] = [float(x) for x in hi]
    return a


def scene_camera_to_perspective(position, rotation_rows, fx, fy, width, height):
    out = Camera()
    pos = (C.c_float * 3)(*[float(x) for x in position])
    rot = (C.c_float * 9)(*[float(x) for row in rotation_rows for x in row])
    _lib.wso_scene_camera_to_perspective(pos, rot, float(fx), float(fy), int(width), int(height), C.byref(out))
    return out


def fit_near_far(cam, aabb):
    _lib.wso_fit_near_far(C.byref(cam), C.byref(aabb))
    return cam


def aabb_radius(aabb):
    return _lib.wso_aabb_radius(C.byref(aabb))


def camera_uniform(cam, vw, vh):
    u = CameraUniform()
    _lib.wso_camera_uniform_build(C.byref(cam), int(vw), int(vh), C.byref(u))
    return u


def settings_uniform(bbox, center, *, gaussian_scaling=1.0, max_sh_deg=3, mip_splatting=None, kernel_size=None,
                     clipping_box=None, walltime=100.0, scene_extend=None, pc_mip=None, pc_kernel_size=None):
    """renderer.rs:620-651 SplattingArgsUniform::from_args_and_pc, restated for the tests."""
    s = SettingsUniform()
    box = clipping_box if clipping_box is not None else bbox
    for i in range(3):
        s.clip_min[i] = box.min[i]
        s.clip_max[i] = box.max[i]
        s.scene_center[i] = center[i]
    s.gaussian_scaling = gaussian_scaling
    s.max_sh_deg = max_sh_deg
    s.mip_splatting = int(mip_splatting if mip_splatting is not None else bool(pc_mip))
    s.kernel_size = kernel_size if kernel_size is not None else (pc_kernel_size if pc_kernel_size is not None else 0.3)
    s.walltime = walltime
    r = aabb_radius(bbox)
    ext = np.float32(scene_extend) if scene_extend is not None else np.float32(r)
    s.scene_extend = float(max(ext, np.float32(r)))
    return s


def ply_rows_convert(rows, sh_deg):
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    n = rows.shape[0]
    g = np.empty((n, 28), dtype=np.uint8)
    s = np.empty((n, 96), dtype=np.uint8)
    _lib.wso_ply_rows_convert(_p(rows), n, int(sh_deg), _p(g), _p(s))
    return g, s


def pointcloud_stats(gaussians, stride, start):
    bbox = Aabb()
    center = (C.c_float * 3)()
    up = (C.c_float * 3)()
    ok = _lib.wso_pointcloud_stats(_p(gaussians), gaussians.shape[0], stride, C.byref(start), C.byref(bbox), center, up)
    return bbox, list(center), (list(up) if ok else None)


def preprocess(gaussians, sh, cam_u, rs_u):
    n = gaussians.shape[0]
    splats = np.empty((n, 20), dtype=np.uint8)
    keys = np.empty(n, dtype=np.uint32)
    src = np.empty(n, dtype=np.uint32)
    v = _lib.wso_preprocess(_p(gaussians), _p(sh), n, C.byref(cam_u), C.byref(rs_u), _p(splats), _p(keys), _p(src))
    return splats[:v].copy(), keys[:v].copy(), src[:v].copy()


def preprocess_compressed(gaussians, sh_bytes, covars, quant, sh_deg, cam_u, rs_u):
    n = gaussians.shape[0]
    splats = np.empty((n, 20), dtype=np.uint8)
    keys = np.empty(n, dtype=np.uint32)
    src = np.empty(n, dtype=np.uint32)
    v = _lib.wso_preprocess_compressed(_p(gaussians), _p(sh_bytes), _p(covars), C.byref(quant), n, int(sh_deg),
                                       C.byref(cam_u), C.byref(rs_u), _p(splats), _p(keys), _p(src))
    return splats[:v].copy(), keys[:v].copy(), src[:v].copy()


def make_quantization(qdict):
    q = GaussianQuantization()
    for name in ("color_dc", "color_rest", "opacity", "scaling_factor"):
        zp, sc = qdict[name]
        getattr(q, name).zero_point = int(zp)
        getattr(q, name).scale = float(sc)
    return q


def sort_pairs(keys, payload):
    k = np.ascontiguousarray(keys, dtype=np.uint32).copy()
    p = np.ascontiguousarray(payload, dtype=np.uint32).copy()
    _lib.wso_sort_pairs(_p(k), _p(p), k.shape[0])
    return k, p


def render(splats, sorted_indices, w, h, background=(0, 0, 0, 0), target_mode=0):
    out = np.empty((h, w, 4), dtype=np.float32)
    bg = (C.c_float * 4)(*[float(x) for x in background])
    splats = np.ascontiguousarray(splats)
    v = splats.shape[0] if sorted_indices is None else len(sorted_indices)
    si = None if sorted_indices is None else np.ascontiguousarray(sorted_indices, dtype=np.uint32)
    _lib.wso_render(_p(splats), _p(si) if si is not None else None, v, int(w), int(h), bg, int(target_mode), _p(out))
    return out


def render_frame(gaussians, sh, cam_u, rs_u, w, h, background=(0, 0, 0, 0), target_mode=0):
    """Whole reference frame on the CPU: K1 -> stable sort by key -> K6."""
    splats, keys, _ = preprocess(gaussians, sh, cam_u, rs_u)
    _, order = sort_pairs(keys, np.arange(len(keys), dtype=np.uint32))
    return render(splats, order, w, h, background, target_mode), len(keys)
