"""Shared scene/camera construction for the parity tests: the same seeded inputs go to the HIP path
(through the C ABI) and to the oracle."""
import numpy as np

from websplat import synth


class Scene:
    def __init__(self, ws, oracle, rows, sh_deg, cam_json, viewport, pc_meta=None, **arg_overrides):
        self.ws, self.oracle = ws, oracle
        self.sh_deg = sh_deg
        self.viewport = (int(viewport[0]), int(viewport[1]))
        self.gpc = ws.GenericGaussianPointCloud.from_ply_rows(rows, sh_deg, **(pc_meta or {}))
        cam = ws.PerspectiveCamera.from_scene_camera(cam_json.position, cam_json.rotation, cam_json.fx, cam_json.fy,
                                                     cam_json.width, cam_json.height)
        cam.fit_near_far(self.gpc.aabb)  # all offline callers do this (bin/render.rs:86, bin/measure.rs:105)
        kw = dict(camera=cam, viewport=self.viewport, max_sh_deg=sh_deg)
        kw.update(arg_overrides)
        self.args = ws.SplattingArgs(**kw)

    def oracle_uniforms(self, pc):
        """Uniforms as the LIBRARY builds them, byte-copied into the oracle's structs: K1 parity is then a
        statement about the kernel alone (uniform parity is tested separately in test_host.py)."""
        cu = self.oracle.copy_struct(self.oracle.CameraUniform, self.args.camera.uniform(self.viewport))
        rs = self.oracle.copy_struct(self.oracle.SettingsUniform, pc.settings_uniform(self.args))
        return cu, rs

    def oracle_k1(self, pc):
        cu, rs = self.oracle_uniforms(pc)
        return self.oracle.preprocess(self.gpc.gaussians, self.gpc.sh_coefs, cu, rs)

    def oracle_image(self, pc, background=(0, 0, 0, 0), target_mode=0):
        splats, keys, src = self.oracle_k1(pc)
        _, order = self.oracle.sort_pairs(keys, np.arange(len(keys), dtype=np.uint32))
        w, h = self.viewport
        return self.oracle.render(splats, order, w, h, background, target_mode), (splats, keys, src, order)


def c1(ws, oracle, n=10_000, viewport=(800, 600), seed=0, sh_deg=3, **kw):
    rows = synth.scene_c1(n=n, seed=seed, sh_deg=sh_deg)
    cj = synth.camera_c1(*viewport)
    cj.fx = cj.fy = float(viewport[0])
    return Scene(ws, oracle, rows, sh_deg, cj, viewport, **kw)


def c2(ws, oracle, n=1_200_000, viewport=(1200, 799), cam_index=0, n_cams=64, seed=1, **kw):
    rows = synth.scene_c2(n=n, seed=seed)
    f = 1200.0 * viewport[0] / 1200.0
    cj = synth.orbit_cameras(n_cams, viewport[0], viewport[1], f, f)[cam_index]
    return Scene(ws, oracle, rows, 3, cj, viewport, **kw)


def half_ulp_diff(a_bits, b_bits):
    """Distance in f16 ulps between two arrays of binary16 bit patterns (monotone integer mapping)."""
    def key(x):
        x = x.astype(np.int32)
        return np.where(x & 0x8000, 0x8000 - (x & 0x7FFF), 0x8000 + (x & 0x7FFF))
    return np.abs(key(a_bits) - key(b_bits))
