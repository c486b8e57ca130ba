"""The seeded inputs of the WGSL golden vectors (tests/golden/wgsl_*.npz): shared by the generator
(tests/golden/gen_wgsl_golden.py, which runs the reference's shader text on them) and by the tests that feed the same
inputs to the oracle and to the HIP path."""
import os
import tempfile

import numpy as np

import scenes
from websplat import synth

K1_CASES = ("default", "sh0", "sh1", "sh2", "mip_on", "kernel_0p1", "kernel_0", "scaling_0p5", "fade_in", "clip_box", "inside",
            "extremes", "planes")
K1C_CASES = ("deg3", "deg2", "deg0", "deg1_mip", "deg3_planes")
FRAME_CASES = ("frame", "frame_opaque")
FRAME_BACKGROUND = {"frame": (0.0, 0.0, 0.0, 0.0), "frame_opaque": (0.25, 0.5, 0.125, 1.0)}

# Cases that keep the identity-rotation camera of synth.camera_c1 (W = transpose(mat3(view)) = I hides a transposed or
# mis-ordered T = W * J, preprocess.wgsl:221-223): only these two.  Every other case looks at the cloud from an oblique
# position, so that every element of the view rotation is non-zero.
IDENTITY_CAMERA = ("sh0", "clip_box")


def oblique_camera(viewport, f):
    return synth.look_at_camera(0, [1.6, -0.9, -2.4], [0.1, 0.05, 0.0], viewport[0], viewport[1], f, f)


def _c1_scene(ws, oracle, name, rows, sh_deg, viewport, **kw):
    if name in IDENTITY_CAMERA:
        cj = synth.camera_c1(*viewport)
        cj.fx = cj.fy = float(viewport[0])
    else:
        cj = oblique_camera(viewport, float(viewport[0]))
    return scenes.Scene(ws, oracle, rows, sh_deg, cj, viewport, max_sh_deg=sh_deg, **kw)


OPAQUE_ROWS = 6


def opaque_rows(rows):
    """The first OPAQUE_ROWS Gaussians become large and fully opaque: opacity logit 14 (sigmoid = 1 - 8e-7, 1.0 as f16) and
    an isotropic scale of 0.09 .. 0.14 (8 .. 14 px at the fixtures' 320x240 camera), spread over the middle of the cloud.
    With sigma >= 8 px the pixel centre nearest to the splat's centre has a = d^2 / (2 sigma^2) < 0.004, i.e.
    exp(-a) * alpha > 0.996: those fragments -- and only those of the whole fixture set -- reach the `min(0.99, .)` of
    gaussian.wgsl:65 (round-3 verdict: moving the clamp to 0.98 left every fixture untouched)."""
    ncol = rows.shape[1]
    k = OPAQUE_ROWS
    rows[:k, ncol - 8] = 14.0
    rows[:k, ncol - 7:ncol - 4] = np.log(np.linspace(0.09, 0.14, k, dtype=np.float32))[:, None]
    rows[:k, 0] = np.linspace(-0.6, 0.6, k, dtype=np.float32)
    rows[:k, 1] = np.array([0.3, -0.25, 0.1, -0.05, 0.35, -0.3], dtype=np.float32)[:k]
    rows[:k, 2] = np.array([0.2, -0.4, 0.6, -0.1, 0.4, 0.0], dtype=np.float32)[:k]
    return rows


def planes_rows(n, seed, sh_deg=3):
    """Gaussians exactly ON the near and the far plane of a camera at the origin looking down +z with znear = 1, zfar = 3
    (proj[2][2] = 1.5, proj[3][2] = -1.5: clip z = 1.5 z - 1.5 is exact, so z / w is exactly 0 and exactly 1), plus a few
    in between and a few just outside.  preprocess.wgsl:190 culls z <= 0 and z >= 1, preprocess_compressed.wgsl:231 only
    z < 0 and z > 1."""
    rng = np.random.default_rng(seed)
    rows = synth.scene_c1(n=n, seed=seed, sh_deg=sh_deg)
    zs = np.array([1.0, 3.0, 1.0, 3.0, 2.0, 1.5, 0.99951171875, 3.001953125], dtype=np.float32)  # (f16-representable)
    rows[:, 2] = zs[np.arange(n) % len(zs)]
    rows[:, 0] = (rng.uniform(-0.3, 0.3, size=n) * rows[:, 2]).astype(np.float16).astype(np.float32)
    rows[:, 1] = (rng.uniform(-0.2, 0.2, size=n) * rows[:, 2]).astype(np.float16).astype(np.float32)
    return rows


def planes_camera(ws, viewport):
    cj = synth.SceneCamera(0, "00000", viewport[0], viewport[1], [0.0, 0.0, 0.0],
                           [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]], float(viewport[0]), float(viewport[0]))
    cam = ws.PerspectiveCamera.from_scene_camera(cj.position, cj.rotation, cj.fx, cj.fy, *viewport)
    cam.znear, cam.zfar = 1.0, 3.0   # NOT fit_near_far: the planes are the point of the case
    return cj, cam


def k1_scene(ws, oracle, name):
    n, viewport, seed, sh_deg, kw = 700, (640, 480), 50, 3, {}
    if name in FRAME_CASES:  # the whole-frame fixtures: the default scene (another seed for the opaque one) on a quarter of the pixels
        if name == "frame_opaque":
            seed, n = 51, 400
        rows = synth.scene_c1(n=n, seed=seed, sh_deg=3)
        if name == "frame_opaque":
            opaque_rows(rows)
        return _c1_scene(ws, oracle, name, rows, 3, (320, 240))
    if name == "planes":
        vp = (400, 300)
        cj, cam = planes_camera(ws, vp)
        sc = scenes.Scene(ws, oracle, planes_rows(64, 77), 3, cj, vp)
        sc.args.camera = cam
        return sc
    if name == "kernel_0":
        # no dilation headroom (kernel_size 0) and sub-pixel scales: mid - radius falls below the 0.1 floor of
        # preprocess.wgsl:246 for most Gaussians (with the default 0.3 px^2 dilation it never does)
        rng = np.random.default_rng(97)
        rows = synth.scene_c1(n=n, seed=96)
        ncol = rows.shape[1]
        rows[:, ncol - 7:ncol - 4] = rng.uniform(np.log(2e-5), np.log(0.02), size=(n, 3)).astype(np.float32)
        return _c1_scene(ws, oracle, name, rows, 3, viewport, kernel_size=0.0)
    if name.startswith("sh"):
        sh_deg = int(name[2])
        seed = 60 + sh_deg
    elif name == "mip_on":
        kw = dict(mip_splatting=True)
    elif name == "kernel_0p1":
        kw = dict(kernel_size=0.1)
    elif name == "scaling_0p5":
        kw = dict(gaussian_scaling=0.5)
    elif name == "fade_in":
        kw = dict(walltime=1.7)
    elif name == "clip_box":
        kw = dict(clipping_box=ws.Aabb([-0.5, -0.25, -1.0], [0.75, 0.5, 0.1]))
    elif name == "inside":  # camera inside the cloud: z <= 0, z >= 1 and the 1.2 w bounds all cull something
        rows = synth.scene_c2(n=n, seed=seed)
        cj = synth.look_at_camera(0, [0.3, -0.2, -1.0], [0.2, 0.1, 0.5], 400, 300, 350.0, 350.0)
        return scenes.Scene(ws, oracle, rows, 3, cj, (400, 300))
    elif name == "extremes":
        # the rarely taken branches: scales from 1e-6 to 3 (lambda2 clamped at 0.1, f16 overflow of the axes to inf for
        # splats the camera almost touches), mip-splatting with det_0 <= 1e-6 (coef = 0), opacities at both ends,
        # Gaussians on the near plane and just outside the 1.2 w cull bounds; camera 0.3 from the cloud's edge
        rng = np.random.default_rng(99)
        rows = synth.scene_c1(n=n, seed=98)
        ncol = rows.shape[1]
        rows[:, ncol - 7:ncol - 4] = rng.uniform(np.log(1e-6), np.log(3.0), size=(n, 3)).astype(np.float32)   # log scales
        rows[:, ncol - 8] = rng.choice(np.array([-30.0, -8.0, 0.0, 8.0, 30.0], dtype=np.float32), size=n)      # opacity logits
        rows[: n // 4, :3] *= np.float32(0.02)                                                                  # a knot at the centre
        cj = synth.look_at_camera(0, [0.0, 0.1, -1.3], [0.0, 0.0, 0.0], 400, 300, 300.0, 300.0)
        return scenes.Scene(ws, oracle, rows, 3, cj, (400, 300), mip_splatting=True, kernel_size=0.05)
    elif name != "default":
        raise KeyError("unknown K1 case " + name)
    return _c1_scene(ws, oracle, name, synth.scene_c1(n=n, seed=seed, sh_deg=sh_deg), sh_deg, viewport, **kw)


def k1c_inputs(ws, name):
    """-> (host point cloud read back through the library's .npz reader, camera, viewport, sh_deg)"""
    sh_deg = {"deg3": 3, "deg2": 2, "deg0": 0, "deg1_mip": 1, "deg3_planes": 3}[name]
    planes = name == "deg3_planes"
    a = synth.c3dgs_arrays(n=64 if planes else 500, n_geometry=96, n_sh=80, seed=(90 if planes else 70) + sh_deg, sh_deg=sh_deg,
                           extent=1.0)
    a["scaling_factor_zero_point"] = np.array(330, dtype=np.int32)
    if planes:  # Gaussians exactly on z / w = 0 and = 1 (kept by `<` / `>` of preprocess_compressed.wgsl:231)
        a["xyz"] = planes_rows(64, 78)[:, :3].astype(np.float16)
    if name == "deg1_mip":  # the file's own metadata (io/npz.rs:29-56): mip-splatting on, kernel size 0.1
        a["mip_splatting"] = np.array(1, dtype=np.int32)
        a["kernel_size"] = np.array(0.1, dtype=np.float32)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "c.npz")
        synth.write_npz(path, a)
        gpc = ws.read_npz(path)
    if planes:
        viewport = (400, 300)
        _, cam = planes_camera(ws, viewport)
        return gpc, cam, viewport, sh_deg
    viewport = (640, 360)
    cj = synth.look_at_camera(0, [0.3, -0.2, -3.0], [0, 0, 0], viewport[0], viewport[1], 700.0, 700.0)
    cam = ws.PerspectiveCamera.from_scene_camera(cj.position, cj.rotation, cj.fx, cj.fy, *viewport)
    cam.fit_near_far(gpc.aabb)
    return gpc, cam, viewport, sh_deg
