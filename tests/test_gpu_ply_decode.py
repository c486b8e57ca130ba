"""SURVEY 8f N1, GPU half: raw PLY vertex rows -> resident scene planes by k_ply_decode (ply_decode.hip), against the
oracle's restatement of io/ply.rs:50-100 (wso_ply_rows_convert) and against the host twin (ws_ply_rows_convert).

Stated tolerance: positions, the SH record and the padding are BYTE-EXACT.  Opacity and the six covariance halves
depend on exp(): the reference calls libm's expf, whose last bit is implementation-defined (glibc's differs from the
correctly rounded value for 0.06 % of arguments); the kernel rounds an f64 exp once.  After the f16 rounding those
seven halves may differ by ONE f16 ulp (off-diagonal covariance elements that cancel to almost nothing: by 4e-7 of the
row's largest element), in at most 0.2 % of the values."""
import os

import numpy as np
import pytest

import scenes
from websplat import synth

pytestmark = pytest.mark.gpu


def _compare(g, s, g0, s0, n):
    assert np.array_equal(g[:, :12], g0[:, :12])                       # x, y, z
    assert np.array_equal(g[:, 14:16], g0[:, 14:16])                   # padding
    assert np.array_equal(s, s0)                                       # SH: pure f32 -> f16 conversions + transpose
    h = np.ascontiguousarray(g[:, 12:28]).view(np.uint16).reshape(n, 8)[:, [0, 2, 3, 4, 5, 6, 7]]
    h0 = np.ascontiguousarray(g0[:, 12:28]).view(np.uint16).reshape(n, 8)[:, [0, 2, 3, 4, 5, 6, 7]]
    ulp = scenes.half_ulp_diff(h, h0)
    # an off-diagonal covariance element is a sum of products of mixed sign: where it cancels to (almost) nothing, the
    # one-f32-ulp difference of a scale is many f16 ulps of the tiny result -- bounded instead by 4e-7 of the row's
    # largest covariance element (an f32 rounding of the terms that cancelled)
    v, v0 = h.view(np.float16).astype(np.float64), h0.view(np.float16).astype(np.float64)
    row_scale = np.abs(v0[:, 1:]).max(axis=1, keepdims=True)
    bad = (ulp > 1) & ~(np.abs(v - v0) <= 4e-7 * row_scale)
    assert not bad.any(), (int(bad.sum()), int(ulp.max()))
    frac = float((ulp != 0).mean())
    assert frac <= 2e-3, frac
    return frac


@pytest.mark.parametrize("sh_deg", [3, 2, 1, 0])
def test_ply_rows_decode_on_gpu_vs_oracle(ws, ctx, oracle, sh_deg):
    n = 50_001
    rows = synth.scene_c1(n=n, seed=70 + sh_deg, sh_deg=sh_deg)
    rng = np.random.default_rng(5)
    tail = 14 + 3 * (sh_deg + 1) ** 2 - 8
    rows[:200, tail] = rng.uniform(-30, 30, 200)           # opacity logits far out on both sigmoid branches
    rows[200:400, tail + 1:tail + 4] = rng.uniform(-14, 3, (200, 3))   # log-scales from 1e-6 to 20
    rows[400:410, tail + 4:tail + 8] *= 1e-3               # tiny (still normalisable) quaternions
    pc = ws.PointCloud.from_ply_rows(ctx, rows, sh_deg, kernel_size=0.125, mip_splatting=False)
    try:
        g, s = pc.download()
        g0, s0 = oracle.ply_rows_convert(rows, sh_deg)
        _compare(g, s, g0, s0, n)
        ref = ws.GenericGaussianPointCloud.from_ply_rows(rows, sh_deg)
        assert np.array_equal(ref.gaussians, g0) and np.array_equal(ref.sh_coefs, s0)      # host twin: byte-exact
        bb = pc.bbox()
        assert np.array_equal(np.float32(list(bb.min)), np.float32(ref.aabb.min))
        assert np.array_equal(np.float32(list(bb.max)), np.float32(ref.aabb.max))
        assert np.array_equal(np.float32(pc.center()), np.float32(ref.center))
        assert pc.dilation_kernel_size() == 0.125 and pc.mip_splatting() is False and pc.sh_deg() == sh_deg and not pc.compressed()
    finally:
        pc.close()


def test_ply_file_gpu_decode_equals_host_decode_image(ws, ctx, oracle, tmp_path):
    """ws_pointcloud_load_ply (GPU decode) and the host conversion give the same picture, and the stated bound holds
    at 1 M rows; the decode rate is printed for the load-time report."""
    import time
    n = 1_000_000
    rows = synth.scene_c2(n=n, seed=1)
    path = str(tmp_path / "scene.ply")
    synth.write_ply(path, rows, 3, comments=["kernel_size=0.3"])
    t0 = time.perf_counter()
    pc_gpu = ws.PointCloud.load(ctx, path)
    t_gpu = time.perf_counter() - t0
    # the host conversion: ws_ply_read + ws_pointcloud_create -- what ws_context_config::ply_decode_host makes ws_pointcloud_load do
    # (the library reads no environment variable, and the switch belongs to the context: nothing to flip on a live one)
    t0 = time.perf_counter()
    pc_host = ws.PointCloud(ctx, ws.read_ply(path))
    t_host = time.perf_counter() - t0
    try:
        g, s = pc_gpu.download()
        g0, s0 = pc_host.download()
        go, so = oracle.ply_rows_convert(rows, 3)
        assert np.array_equal(g0, go) and np.array_equal(s0, so)      # host path: byte-exact vs the oracle
        frac = _compare(g, s, g0, s0, n)
        print(f"\\nPLY load, {n} rows (248 MB): GPU decode {t_gpu * 1e3:.0f} ms, host decode {t_host * 1e3:.0f} ms; "
              f"f16 halves differing by one ulp: {frac:.2e}")
        assert pc_gpu.dilation_kernel_size() == pytest.approx(0.3)
        cj = synth.orbit_cameras(8, 960, 540, 960.0, 960.0)[3]
        cam = ws.PerspectiveCamera.from_scene_camera(cj.position, cj.rotation, cj.fx, cj.fy, 960, 540)
        cam.fit_near_far(pc_gpu.bbox())
        args = ws.SplattingArgs(camera=cam, viewport=(960, 540), max_sh_deg=3)
        r = ws.GaussianRenderer(ctx, "rgba32float", 3, False)
        imgs = []
        for pc in (pc_gpu, pc_host):
            r.prepare(pc, args)
            r.render(pc)
            imgs.append(r.download_target())
            assert r.errors()[0] == 0
        r.close()
        ok, msg, *_ = scenes.image_close(imgs[0], imgs[1])
        assert ok, msg
    finally:
        pc_gpu.close()
        pc_host.close()
