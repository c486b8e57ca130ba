"""The seeded inputs of the WGSL golden vectors (tests/golden/wgsl_*.npz): shared by the generator
(tests/golden/gen_wgsl_golden.py, which runs the reference's shader text on them) and by the tests that feed the same
inputs to the oracle and to the HIP path."""
import os
import tempfile

import numpy as np

import scenes
from websplat import synth

K1_CASES = ("default", "sh0", "sh1", "sh2", "mip_on", "kernel_0p1", "scaling_0p5", "fade_in", "clip_box", "inside", "extremes")
K1C_CASES = ("deg3", "deg2", "deg0", "deg1_mip")


def k1_scene(ws, oracle, name):
    n, viewport, seed, sh_deg, kw = 700, (640, 480), 50, 3, {}
    if name == "frame":  # the whole-frame fixture: the default scene on a quarter of the pixels
        return scenes.c1(ws, oracle, n=n, viewport=(320, 240), seed=seed, sh_deg=3, max_sh_deg=3)
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
    return scenes.c1(ws, oracle, n=n, viewport=viewport, seed=seed, sh_deg=sh_deg, max_sh_deg=sh_deg, **kw)


def k1c_inputs(ws, name):
    """-> (host point cloud read back through the library's .npz reader, camera, viewport, sh_deg)"""
    sh_deg = {"deg3": 3, "deg2": 2, "deg0": 0, "deg1_mip": 1}[name]
    a = synth.c3dgs_arrays(n=500, n_geometry=96, n_sh=80, seed=70 + sh_deg, sh_deg=sh_deg, extent=1.0)
    a["scaling_factor_zero_point"] = np.array(330, dtype=np.int32)
    if name == "deg1_mip":  # the file's own metadata (io/npz.rs:29-56): mip-splatting on, kernel size 0.1
        a["mip_splatting"] = np.array(1, dtype=np.int32)
        a["kernel_size"] = np.array(0.1, dtype=np.float32)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "c.npz")
        synth.write_npz(path, a)
        gpc = ws.read_npz(path)
    viewport = (640, 360)
    cj = synth.look_at_camera(0, [0.3, -0.2, -3.0], [0, 0, 0], viewport[0], viewport[1], 700.0, 700.0)
    cam = ws.PerspectiveCamera.from_scene_camera(cj.position, cj.rotation, cj.fx, cj.fy, *viewport)
    cam.fit_near_far(gpc.aabb)
    return gpc, cam, viewport, sh_deg
