"""GPU parity of the whole render path (prepare + render through the C ABI) against the oracle image
(K1 -> stable depth sort -> gaussian.wgsl quad rasterisation + PREMULTIPLIED_ALPHA_BLENDING).

Stated tolerance (SURVEY 8c / BASELINE north star "within a stated float tolerance"), defined in scenes.py:
  premultiplied RGBA, f32 target:  max-abs <= 2e-3, mean-abs <= 1e-4  vs the oracle's f32-target image, except
  for at most max(4, 2e-5 * pixels) cut-off-boundary pixels, each within ONE boundary fragment (0.0135).
Sources of the gap: front-to-back vs back-to-front summation order, the T < 2^-14 early-out, device exp, and the
last-ulp difference in `a` next to the discard threshold of gaussian.wgsl:61."""
import numpy as np
import pytest

from variants import env_param, exp_param  # noqa: F401

import scenes
from websplat import synth

pytestmark = pytest.mark.gpu

MAX_ABS = scenes.MAX_ABS
MEAN_ABS = scenes.MEAN_ABS


def _render(ws, ctx, scene, fmt="rgba32float", background=(0, 0, 0, 0), pc=None, sh_deg=None):
    own = pc is None
    if own:
        pc = ws.PointCloud(ctx, scene.gpc)
    r = ws.GaussianRenderer(ctx, fmt, scene.sh_deg if sh_deg is None else sh_deg, False)
    try:
        r.prepare(pc, scene.args)
        r.render(pc, background=background)
        img = r.download_target()
        stats = r.frame_stats()
    finally:
        r.close()
    return pc, img, stats


def _assert_close(img, ref, max_abs=MAX_ABS, mean_abs=MEAN_ABS, proof=None):
    # tighter-than-default bounds (analytic single-splat tests) do not get the boundary allowance; where it applies, every
    # pixel that uses it has to be PROVEN a cut-off boundary pixel (scenes.BoundaryProof)
    ok, msg, mx, mean, _ = scenes.image_close(img, ref, max_abs, mean_abs, allow_boundary=max_abs >= MAX_ABS, proof=proof)
    assert ok, msg
    return mx, mean


def test_image_c1(ws, ctx, oracle):
    """BASELINE config 1: 10 k Gaussians, 800x600, single view."""
    sc = scenes.c1(ws, oracle)
    pc, img, stats = _render(ws, ctx, sc)
    try:
        ref, ofr_ = sc.oracle_image(pc)
        assert ref[..., 3].max() > 0.9 and (ref[..., 3] > 0).mean() > 0.2  # the scene actually covers the image
        _assert_close(img, ref, proof=sc.proof(ofr_))
        assert stats["overflow"] == 0
    finally:
        pc.close()


@pytest.mark.parametrize("viewport", [(801, 599), (16, 16), (17, 33), (250, 7), (4096, 4096)])
def test_image_odd_viewports(ws, ctx, oracle, viewport):
    """Viewports that are not multiples of the tile, down to a single tile (one sort pass, ranges written by that pass),
    up to 16384 tiles of 32x32 px (two 7-bit passes); test_many_tiles_32bit_keys covers the 65536-tile case."""
    sc = scenes.c1(ws, oracle, n=3000, viewport=viewport, seed=9)
    pc, img, _ = _render(ws, ctx, sc)
    try:
        ref, ofr_ = sc.oracle_image(pc)
        assert img.shape == (viewport[1], viewport[0], 4)
        _assert_close(img, ref, proof=sc.proof(ofr_))
    finally:
        pc.close()


def test_image_background_and_formats(ws, ctx, oracle):
    """Clear colour (bin/render.rs:113-116 uses TRANSPARENT, the viewer a user colour) and the three target
    formats the reference's front-ends use.  f16 / unorm8 targets are rounded once at the end here, whereas
    the reference's ROPs round after every blend: compare against the f32 oracle with the format's own
    quantisation step as tolerance."""
    sc = scenes.c1(ws, oracle, n=5000, viewport=(320, 240), seed=12)
    bg = (0.1, 0.3, 0.5, 1.0)
    pc, img32, _ = _render(ws, ctx, sc, background=bg)
    try:
        ref, ofr_ = sc.oracle_image(pc, background=bg)
        _assert_close(img32, ref, proof=sc.proof(ofr_, bg))
        assert np.allclose(img32[0, 0], bg, atol=1e-6) or ref[0, 0, 3] != 1.0
        _, img16, _ = _render(ws, ctx, sc, fmt="rgba16float", background=bg, pc=pc)
        assert img16.dtype == np.float16
        ok, msg, *_ = scenes.image_close(img16.astype(np.float32), ref, max_abs=MAX_ABS + 2e-3, mean_abs=1e-3)
        assert ok, msg  # f16 ulp at ~2.0 is 2e-3
        _, img8, _ = _render(ws, ctx, sc, fmt="rgba8unorm", background=bg, pc=pc)
        assert img8.dtype == np.uint8
        want8 = np.clip(ref, 0, 1) * 255.0
        ok, msg, *_ = scenes.image_close(img8.astype(np.float32) / 255.0, want8 / 255.0,
                                         max_abs=(0.5 + 255 * MAX_ABS + 1e-3) / 255.0, mean_abs=0.5 / 255.0)
        assert ok, msg
    finally:
        pc.close()


def test_single_splat_analytic(ws, ctx, oracle):
    """gaussian.wgsl:59-67 on one isotropic Gaussian straight ahead: alpha(x) = min(0.99, a * exp(-r^2/2s^2))
    inside a <= 2*CUTOFF, exactly 0 outside."""
    # The Gaussian sits away from the origin: the PLY bbox starts at the origin (Aabb::zeroed, io/mod.rs:74), so
    # the bbox radius (= scene_extend) is > 0 and the fade-in term (preprocess.wgsl:196-203) is 1 at walltime 100.
    row = np.zeros((1, 62), dtype=np.float32)
    row[0, 0:3] = [0.5, 0.0, 0.0]
    row[0, 6:9] = [1.0, 0.5, -0.5]
    row[0, 54] = 14.0                     # opacity logit: sigmoid = 1 - 8e-7, alpha = 1.0 as f16 -> the four pixel centres
                                          # around the splat's centre have exp(-a) * alpha > 0.99: the clamp is reached
    row[0, 55:58] = np.log(0.05)          # isotropic scale
    row[0, 58:62] = [1, 0, 0, 0]
    cj = synth.look_at_camera(0, [0.5, 0.0, -2.0], [0.5, 0, 0], 128, 128, 256.0, 256.0)
    sc = scenes.Scene(ws, oracle, row, 3, cj, (128, 128))
    pc, img, stats = _render(ws, ctx, sc)
    try:
        assert stats["num_visible"] == 1
        ref, (splats, _, _, _) = sc.oracle_image(pc)
        _assert_close(img, ref, max_abs=5e-6, mean_abs=1e-6)
        h = splats.view(np.uint16).reshape(10)
        f = [oracle.f16_to_f32(int(x)) for x in h]
        # axes in px (|v| = sqrt(2 lambda)), centre px; kept region a = |M^-1 d|^2 <= 2*CUTOFF
        M = np.array([[f[0] * 128, f[2] * 128], [-f[1] * 128, -f[3] * 128]])
        c = np.array([(f[4] * 0.5 + 0.5) * 128, (0.5 - f[5] * 0.5) * 128])
        ys, xs = np.mgrid[0:128, 0:128]
        d = np.stack([xs + 0.5 - c[0], ys + 0.5 - c[1]], -1)
        p = d @ np.linalg.inv(M).T
        a = (p ** 2).sum(-1)
        want = np.where(a <= 2 * 2.3539888583335364, np.minimum(0.99, np.exp(-a) * f[9]), 0.0)
        assert np.abs(img[..., 3] - want).max() < 1e-5
        assert (img[..., 3] == 0).sum() == (want == 0).sum()
        assert f[9] == 1.0 and (want == 0.99).sum() >= 4 and (img[..., 3] == np.float32(0.99)).sum() == (want == 0.99).sum()
        lam = 0.5 * (M[0, 0] ** 2 + M[1, 0] ** 2)  # eigenvalue of the screen covariance
        sigma_px = 0.05 * 256.0 / 2.0              # sigma * f / z
        assert np.isclose(lam, sigma_px ** 2 + 0.3, rtol=2e-3)  # + dilation kernel (preprocess.wgsl:238-240)
    finally:
        pc.close()


def test_two_splats_depth_order(ws, ctx, oracle):
    """Blend order follows depth, not storage order (renderer.rs:65 + key order preprocess.wgsl:273)."""
    def rows(order):
        # two invisible (opacity ~ 0) Gaussians at z = +-2 stretch the bbox so that fit_near_far does not put
        # the near plane exactly on the near splat (z <= 0 is culled, preprocess.wgsl:190)
        r = np.zeros((4, 62), dtype=np.float32)
        r[2, 0:3], r[3, 0:3] = [0.0, 0.0, -2.0], [0.0, 0.0, 2.0]
        r[2:, 54] = -30.0
        r[2:, 55:58] = np.log(0.01)
        r[2:, 58] = 1.0
        near = dict(z=-0.5, col=[3.0, -3.0, -3.0])
        far = dict(z=0.5, col=[-3.0, -3.0, 3.0])
        for i, s in enumerate(order):
            g = near if s == "near" else far
            r[i, 0:3] = [0.0, 0.0, g["z"]]
            r[i, 6:9] = g["col"]
            r[i, 54] = 8.0
            r[i, 55:58] = np.log(0.2)
            r[i, 58:62] = [1, 0, 0, 0]
        return r
    cj = synth.look_at_camera(0, [0.0, 0.0, -4.0], [0, 0, 0], 96, 96, 160.0, 160.0)
    imgs = []
    for order in (("near", "far"), ("far", "near")):
        sc = scenes.Scene(ws, oracle, rows(order), 3, cj, (96, 96))
        pc, img, _ = _render(ws, ctx, sc)
        ref, _ = sc.oracle_image(pc)
        _assert_close(img, ref, max_abs=1e-5, mean_abs=1e-6)
        imgs.append(img)
        pc.close()
    assert np.array_equal(imgs[0], imgs[1])
    centre = imgs[0][48, 48]
    assert centre[0] > 0.9 and centre[2] < 0.02  # the near (red) splat dominates


def test_determinism_and_reuse(ws, ctx, oracle):
    """Same view twice, and after rendering another view in between: identical bytes (ordered compaction +
    stable sorts make the frame a pure function of scene and camera)."""
    rows = synth.scene_c2(n=80_000, seed=6)
    cams = synth.orbit_cameras(4, 480, 320, 420.0, 420.0)
    sc0 = scenes.Scene(ws, oracle, rows, 3, cams[0], (480, 320))
    sc1 = scenes.Scene(ws, oracle, rows, 3, cams[1], (480, 320))
    pc = ws.PointCloud(ctx, sc0.gpc)
    r = ws.GaussianRenderer(ctx, "rgba32float", 3, False)
    r2 = ws.GaussianRenderer(ctx, "rgba32float", 3, False)
    try:
        out = []
        for sc in (sc0, sc1, sc0):
            r.prepare(pc, sc.args)
            r.render(pc)
            out.append(r.download_target().copy())
        assert np.array_equal(out[0], out[2])
        assert not np.array_equal(out[0], out[1])
        r2.prepare(pc, sc0.args)  # a second renderer on the same point cloud: private scratch
        r2.render(pc)
        assert np.array_equal(r2.download_target(), out[0])
        ref, ofr_ = sc0.oracle_image(pc)
        _assert_close(out[0], ref, proof=sc0.proof(ofr_))
    finally:
        r.close()
        r2.close()
        pc.close()


def test_image_c2_subsample(ws, ctx, oracle):
    """Bonsai-like scene with deep overdraw (300 k Gaussians at 1200x799): early-out and ordering under load."""
    sc = scenes.c2(ws, oracle, n=300_000)
    pc, img, stats = _render(ws, ctx, sc)
    try:
        ref, ofr_ = sc.oracle_image(pc)
        mx, mean = _assert_close(img, ref, proof=sc.proof(ofr_))
        assert stats["overflow"] == 0
        assert stats["num_tile_entries"] > stats["num_visible"]
    finally:
        pc.close()


def test_errors_and_capacity(ws, ctx, oracle):
    sc = scenes.c1(ws, oracle, n=4000, viewport=(320, 240), seed=2)
    pc = ws.PointCloud(ctx, sc.gpc)
    r = ws.GaussianRenderer(ctx, "rgba32float", 3, False)
    try:
        with pytest.raises(ws.WebSplatError) as e:  # render before prepare (the reference unwrap()s here)
            r._viewport = (320, 240)
            r.render(pc)
        assert e.value.code == -5
        r.set_tile_entry_capacity(1000)  # far too small: entries are dropped, flagged, nothing crashes
        r.prepare(pc, sc.args)
        r.render(pc)
        st = r.frame_stats()
        assert st["overflow"] & 1 and st["num_tile_entries"] == 1000
        r.set_tile_entry_capacity(0)
        r.prepare(pc, sc.args)
        r.render(pc)
        assert r.frame_stats()["overflow"] == 0
        with pytest.raises(ws.WebSplatError):
            ws.GaussianRenderer(ctx, "rgba32float", 4, False)  # sh_deg > 3
        bad = ws.SplattingArgs(camera=sc.args.camera, viewport=(0, 10))
        with pytest.raises(ws.WebSplatError):
            r.prepare(pc, bad)
    finally:
        r.close()
        pc.close()


def test_ply_file_roundtrip(ws, ctx, oracle, tmp_path):
    """io/ply.rs through the library's own loader: header comments, little and big endian bodies."""
    rows = synth.scene_c1(n=2000, seed=8)
    for big in (False, True):
        p = tmp_path / f"scene_{int(big)}.ply"
        synth.write_ply(str(p), rows, 3, comments=["mip=true", "kernel_size=0.25", "background_color=0.1,0.2,0.3"],
                        big_endian=big)
        pc = ws.PointCloud.load_ply(ctx, str(p))
        try:
            assert pc.num_points() == 2000 and pc.sh_deg() == 3 and not pc.compressed()
            assert pc.mip_splatting() is True
            assert np.isclose(pc.dilation_kernel_size(), 0.25)
            assert np.allclose(pc.background_color(), [0.1, 0.2, 0.3])
            g, _ = oracle.ply_rows_convert(rows, 3)
            bbox, center, _ = oracle.pointcloud_stats(g, 28, oracle.make_aabb([0, 0, 0], [0, 0, 0]))
            assert np.allclose(pc.bbox().min, list(bbox.min)) and np.allclose(pc.bbox().max, list(bbox.max))
            assert np.allclose(pc.center(), center)
            sc = scenes.c1(ws, oracle, n=2000, viewport=(200, 150), seed=8,
                           pc_meta=dict(mip_splatting=True, kernel_size=0.25))
            _, img, _ = _render(ws, ctx, sc, pc=pc)
            ref, ofr_ = sc.oracle_image(pc)
            _assert_close(img, ref, proof=sc.proof(ofr_))
        finally:
            pc.close()


@pytest.mark.parametrize("env", [env_param({"WS_BLEND_VARIANT": "1"}), env_param({"WS_DEPTH_SORT": "onesweep", "WS_BLEND_VARIANT": "1"}),
                                 env_param({"WS_TILE_SHAPE": "2x2"}), env_param({"WS_TILE_SHAPE": "4x2"}), env_param({"WS_TILE_SHAPE": "4x4"}),
                                 env_param({"WS_TILE_SHAPE": "4x2", "WS_BLEND_VARIANT": "1"}),
                                 env_param({"WS_TILE_SHAPE": "4x4", "WS_BLEND_VARIANT": "1", "WS_DEPTH_SORT": "onesweep"}),
                                 env_param({"WS_TILE_SHAPE": "4x4", "WS_BLEND_TPW_LOG2": "1"}),
                                 env_param({"WS_TILE_SHAPE": "4x2", "WS_BLEND_TPW_LOG2": "2"}),
                                 env_param({"WS_DEPTH_SORT": "onesweep", "WS_TILE_SHAPE": "2x2"}),
                                 env_param({"WS_BLEND_SPLIT": "1"}), env_param({"WS_BLEND_SPLIT": "0"}),
                                 env_param({"WS_BLEND_DMA": "1"}), env_param({"WS_BLEND_DMA": "1", "WS_BLEND_TPW_LOG2": "2"}),
                                 env_param({"WS_DEPTH_SORT": "onesweep"}), env_param({"WS_DEPTH_SORT": "coop"}), env_param({"WS_DEPTH_SORT": "scan"})])
def test_cross_check_paths(ws, oracle, env, monkeypatch):
    """The alternative implementations kept as cross-checks (fat-tile one-sweep depth sort as per-pass launches and as one
    launch with device-wide barriers, wave-per-quadrant blend, LDS-DMA staging) and
    every tile shape (16x16, 32x16, 32x32 binning tiles) must give the same image as the oracle: a context reads the
    selection from the environment when it is created."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    c = ws.Context(0)
    try:
        sc = scenes.c1(ws, oracle, n=20_000, viewport=(640, 480), seed=12)
        pc, img, stats = _render(ws, c, sc)
        try:
            ref, ofr_ = sc.oracle_image(pc)
            _assert_close(img, ref, proof=sc.proof(ofr_))
            assert stats["overflow"] == 0
        finally:
            pc.close()
    finally:
        c.close()


@pytest.mark.parametrize("env", [env_param({"WS_BLEND_DMA": "1"}), env_param({"WS_BLEND_DMA": "1", "WS_BLEND_TPW_LOG2": "2"}),
                                 env_param({"WS_DEPTH_SORT": "onesweep"}), env_param({"WS_DEPTH_SORT": "coop"}), env_param({"WS_DEPTH_SORT": "scan"})])
@pytest.mark.parametrize("kind", ["c2", "c3"])
def test_switched_paths_draw_the_default_image_bit_for_bit(ws, ctx, oracle, env, kind, monkeypatch):
    """Switches that change HOW a frame is computed, not WHAT (the LDS-DMA staging of the blend had no test; the
    depth-sort forms): the image, the visible count and the entry count equal the default path's bit for bit, on a scene
    of multi-tile splats and on one of pixel-sized splats, and again for a second render() of the same prepared frame
    (the staging buffers / status words of the first are reused)."""
    if kind == "c2":
        rows, viewport = synth.scene_c2(n=200_000, seed=21), (1283, 721)
        cj = synth.orbit_cameras(8, viewport[0], viewport[1], 900.0, 900.0)[3]
    else:
        rows, viewport = synth.scene_c3(n=300_000, seed=22), (640, 480)
        cj = synth.look_at_camera(0, [0.0, 0.0, -9.0], [0, 0, 0], viewport[0], viewport[1], 520.0, 520.0)
    sc = scenes.Scene(ws, oracle, rows, 3, cj, viewport)
    pc, want, st0 = _render(ws, ctx, sc)
    pc.close()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    c = ws.Context(0)
    try:
        pc = ws.PointCloud(c, sc.gpc)
        r = ws.GaussianRenderer(c, "rgba32float", 3, False)
        try:
            for frame in range(2):
                r.prepare(pc, sc.args)
                r.render(pc)
                img = r.download_target()
                st = r.frame_stats()
                assert st["num_visible"] == st0["num_visible"] and st["num_tile_entries"] == st0["num_tile_entries"]
                assert st["overflow"] == 0 and r.errors()[0] == 0
                assert np.array_equal(img, want), (env, kind, frame, float(np.abs(img - want).max()))
                r.render(pc)   # the same prepared frame once more
                assert np.array_equal(r.download_target(), want), (env, kind, frame, "second render")
        finally:
            r.close()
            pc.close()
    finally:
        c.close()


@pytest.mark.parametrize("env", [env_param({}), env_param({"WS_DEPTH_SORT": "onesweep"}), env_param({"WS_DEPTH_SORT": "coop"}), env_param({"WS_TILE_SORT": "wide"}),
                                 env_param({"WS_BLEND_VARIANT": "1"})])
def test_degenerate_frames(ws, oracle, env, monkeypatch):
    """The ragged ends of the frame: a camera that looks AWAY from the cloud (nothing visible: every kernel behind K1 runs on a
    device-side count of zero), a one-Gaussian cloud, and a cloud whose visible splats all sit in the 1.2x cull margin with
    their footprints off screen (V > 0, no tile entry at all).  The image is the clear colour exactly where nothing is drawn,
    the counters say so, no error bit is set, and a normal frame on the same renderer afterwards is unharmed -- in every
    depth-sort / tile-sort / blend form."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    c = ws.Context(0)
    bg = (0.25, 0.5, 0.125, 1.0)
    try:
        sc = scenes.c1(ws, oracle, n=5000, viewport=(320, 240), seed=31)
        pc = ws.PointCloud(c, sc.gpc)
        r = ws.GaussianRenderer(c, "rgba32float", 3, False)
        try:
            import dataclasses
            # 1. looking away: the same camera turned by 180 degrees about the y axis
            cam = sc.args.camera
            away_cj = synth.look_at_camera(0, [0.0, 0.0, -3.0], [0.0, 0.0, -9.0], 320, 240, 320.0, 320.0)
            away = ws.PerspectiveCamera.from_scene_camera(away_cj.position, away_cj.rotation, away_cj.fx, away_cj.fy, 320, 240)
            away.znear, away.zfar = cam.znear, cam.zfar
            args_away = dataclasses.replace(sc.args, camera=away)
            r.prepare(pc, args_away)
            r.render(pc, background=bg)
            img = r.download_target()
            st = r.frame_stats()
            assert st["num_visible"] == 0 and st["num_tile_entries"] == 0 and st["overflow"] == 0 and r.errors()[0] == 0
            assert np.array_equal(img, np.broadcast_to(np.array(bg, dtype=np.float32), img.shape))
            # 2. a normal frame afterwards, then the empty one again (scratch and epochs are reused)
            r.prepare(pc, sc.args)
            r.render(pc, background=bg)
            good = r.download_target()
            ref, ofr_ = sc.oracle_image(pc, background=bg)
            _assert_close(good, ref, proof=sc.proof(ofr_, bg))
            r.prepare(pc, args_away)
            r.render(pc, background=bg)
            assert np.array_equal(r.download_target(), img)
        finally:
            r.close()
            pc.close()
        # 3. one Gaussian
        row = np.zeros((1, 62), dtype=np.float32)
        row[0, 0:3] = [0.5, 0.0, 0.0]
        row[0, 6:9] = [1.0, 0.5, -0.5]
        row[0, 54] = 2.0
        row[0, 55:58] = np.log(0.05)
        row[0, 58:62] = [1, 0, 0, 0]
        cj = synth.look_at_camera(0, [0.5, 0.0, -2.0], [0.5, 0, 0], 128, 128, 256.0, 256.0)
        one = scenes.Scene(ws, oracle, row, 3, cj, (128, 128))
        pc1, img1, st1 = _render(ws, c, one, background=bg)
        try:
            assert st1["num_visible"] == 1 and st1["num_tile_entries"] >= 1
            ref1, ofr1 = one.oracle_image(pc1, background=bg)
            _assert_close(img1, ref1, proof=one.proof(ofr1, bg))
        finally:
            pc1.close()
        # 4. visible by the cull test (centre within 1.2 x the clip bounds) but every footprint off screen: small splats just
        #    beyond the right edge of the image
        rows = synth.scene_c1(n=2000, seed=32)
        rows[:, 0] = 1.62 + 0.05 * rows[:, 0]     # x: a slab at 1.08..1.17 x the half-width at the cloud's depth (z = 0, camera at -3, tan = 0.5)
        rows[:, 1] *= 0.5
        rows[:, 2] *= 0.05
        rows[:, 55:58] = np.log(0.002)
        cj = synth.camera_c1(320, 240)
        cj.fx = cj.fy = 320.0
        edge = scenes.Scene(ws, oracle, rows, 3, cj, (320, 240))
        pce, imge, ste = _render(ws, c, edge, background=bg)
        try:
            assert ste["overflow"] == 0
            refe, ofre = edge.oracle_image(pce, background=bg)
            assert ste["num_visible"] == len(ofre[1])            # the cull test agrees with the oracle's
            if ste["num_visible"] > 0 and ste["num_tile_entries"] == 0:
                assert np.array_equal(imge, np.broadcast_to(np.array(bg, dtype=np.float32), imge.shape))
            _assert_close(imge, refe, proof=edge.proof(ofre, bg))
        finally:
            pce.close()
    finally:
        c.close()


def test_many_tiles_32bit_keys(ws, oracle, monkeypatch):
    """65536 tiles (4096x4096 at 16x16): the tile sort switches from 16-bit to 32-bit tile ids, two 8-bit passes."""
    monkeypatch.setenv("WS_TILE_SHAPE", "2x2")
    c = ws.Context(0)
    try:
        assert c.tile_size() == (16, 16)
        sc = scenes.c1(ws, oracle, n=3000, viewport=(4096, 4096), seed=9)
        pc, img, stats = _render(ws, c, sc)
        try:
            ref, ofr_ = sc.oracle_image(pc)
            _assert_close(img, ref, proof=sc.proof(ofr_))
            assert stats["overflow"] == 0
        finally:
            pc.close()
    finally:
        c.close()


@pytest.mark.parametrize("shape", ["2x2", "4x2", "4x4"])
@pytest.mark.parametrize("viewport", [(801, 599), (17, 33), (250, 7), (1283, 721)])
def test_tile_shapes_odd_viewports(ws, oracle, shape, viewport, monkeypatch):
    """Every binning-tile shape on viewports that are not multiples of the tile (partial quadrants, partial 64-px
    blocks, a single tile)."""
    monkeypatch.setenv("WS_TILE_SHAPE", shape)
    c = ws.Context(0)
    try:
        sc = scenes.c1(ws, oracle, n=4000, viewport=viewport, seed=13)
        pc, img, stats = _render(ws, c, sc)
        try:
            ref, ofr_ = sc.oracle_image(pc)
            _assert_close(img, ref, proof=sc.proof(ofr_))
            assert stats["overflow"] == 0
        finally:
            pc.close()
    finally:
        c.close()


def _coverage_tiles(splats_f16, viewport, tile=(16, 16)):
    """For every splat: the set of tiles holding at least one pixel centre with a <= 2*CUTOFF, evaluated as the
    oracle does (oracle/ws_oracle.c setup_splat + wso_render), in float64 with a small slack towards 'covered'."""
    w, h = viewport
    tw, th = tile
    tx_n = (w + tw - 1) // tw
    out = []
    cut = 2 * 2.3539888583335364
    for s in splats_f16.astype(np.float64):
        m00, m10, m01, m11 = s[0] * w, -s[1] * h, s[2] * w, -s[3] * h
        det = m00 * m11 - m01 * m10
        tiles = set()
        if det != 0 and np.isfinite(det):
            cx, cy = (s[4] * 0.5 + 0.5) * w, (0.5 - s[5] * 0.5) * h
            i00, i01, i10, i11 = m11 / det, -m01 / det, -m10 / det, m00 / det
            ex = np.sqrt(cut) * np.hypot(m00, m01) + 1.0
            ey = np.sqrt(cut) * np.hypot(m10, m11) + 1.0
            x0, x1 = max(int(np.floor(cx - ex)), 0), min(int(np.ceil(cx + ex)), w - 1)
            y0, y1 = max(int(np.floor(cy - ey)), 0), min(int(np.ceil(cy + ey)), h - 1)
            if x0 <= x1 and y0 <= y1:
                xs = np.arange(x0, x1 + 1) + 0.5 - cx
                ys = np.arange(y0, y1 + 1) + 0.5 - cy
                dx, dy = np.meshgrid(xs, ys)
                a = (i00 * dx + i01 * dy) ** 2 + (i10 * dx + i11 * dy) ** 2
                yy, xx = np.nonzero(a <= cut * (1 - 1e-6))
                tiles = set(((yy + y0) // th * tx_n + (xx + x0) // tw).tolist())
        out.append(tiles)
    return out


@pytest.mark.parametrize("footprint", ["rect", exp_param("ellipse"), "wide"])
@pytest.mark.parametrize("shape", [None, "2x2", "4x2", "4x4"])
@pytest.mark.parametrize("kind", ["c1", "needles"])
def test_binning_covers_every_touched_tile(ws, oracle, kind, shape, footprint, monkeypatch):
    """Binning hands the blend the tiles of the kept ellipse's bounding rectangle (default; "wide": the same rectangle
    through the tile-count form that viewports beyond 256 tiles per axis take) or, with WS_FOOTPRINT=ellipse, the tiles
    the ellipse itself reaches (footprint.h: per tile row the exact column span).  It may list a tile the ellipse misses
    (the blend's exact per-quadrant test drops it) but never drop one it touches: every tile holding a covered pixel
    centre must list the splat, exactly once, and each tile's list must be in draw order."""
    if shape:
        monkeypatch.setenv("WS_TILE_SHAPE", shape)
    if footprint == "ellipse":
        monkeypatch.setenv("WS_FOOTPRINT", "ellipse")
    ctx = ws.Context(0)
    tile = ctx.tile_size()
    if shape:
        assert tile == (8 * int(shape[0]), 8 * int(shape[2]))
    if kind == "c1":
        rows = synth.scene_c1(n=6000, seed=21)
    else:  # long thin splats at all orientations: the case a bounding rectangle is worst at
        rng = np.random.default_rng(22)
        rows = synth.scene_c1(n=3000, seed=22)
        rows[:, -7:-4] = np.log(np.stack([rng.uniform(0.1, 0.4, 3000), rng.uniform(0.003, 0.01, 3000),
                                          rng.uniform(0.003, 0.01, 3000)], 1)).astype(np.float32)
    # "wide": more than 256 binning tiles per axis (even at 32-px tiles) -> the footprint word is a tile count
    viewport = (8256, 80) if footprint == "wide" else (640, 400)
    gpc = ws.GenericGaussianPointCloud.from_ply_rows(rows, 3)
    cj = synth.camera_c1(*viewport)
    cj.fx = cj.fy = 6000.0 if footprint == "wide" else 600.0
    cam = ws.PerspectiveCamera.from_scene_camera(cj.position, cj.rotation, cj.fx, cj.fy, *viewport)
    cam.fit_near_far(gpc.aabb)
    args = ws.SplattingArgs(camera=cam, viewport=viewport, max_sh_deg=3)
    pc = ws.PointCloud(ctx, gpc)
    r = ws.GaussianRenderer(ctx, "rgba32float", 3, False)
    try:
        r.prepare(pc, args)
        r.render(pc)
        st = r.frame_stats()
        assert st["overflow"] == 0
        fr = r.download_frame()
        # the device may have binned this frame at 2 x 2 compositing tiles (ws_renderer_binning_tile): the lists, and the
        # tiles a splat has to be listed in, are per BINNING tile
        bt = r.binning_tile()
        assert bt in (tile, (2 * tile[0], 2 * tile[1])) and (bt == tile or (shape in (None, "4x4") and footprint == "rect"))
        tile = bt
        begin, end, entries = r.tile_lists()
        assert len(entries) == st["num_tile_entries"] == int((end - begin).sum())
        rank = np.empty(st["num_visible"], dtype=np.int64)
        rank[fr["sorted"]] = np.arange(st["num_visible"])
        listed = [set() for _ in range(st["num_visible"])]
        for t in range(len(begin)):
            e = entries[begin[t]:end[t]]
            assert len(np.unique(e)) == len(e)                        # a splat appears once per tile
            assert np.all(np.diff(rank[e]) > 0)                        # far -> near inside the tile
            for sidx in e.tolist():
                listed[sidx].add(t)
        assert len(begin) == -(-viewport[0] // tile[0]) * -(-viewport[1] // tile[1])
        cov = _coverage_tiles(fr["splats"].view(np.float16).reshape(-1, 10), viewport, tile)
        missing = [(i, sorted(c - listed[i])) for i, c in enumerate(cov) if not c <= listed[i]]
        assert not missing, missing[:5]
        n_cov, n_listed = sum(len(c) for c in cov), sum(len(l) for l in listed)
        if footprint == "ellipse":
            assert n_listed <= 1.05 * n_cov + 50, (n_listed, n_cov)  # the ellipse's footprint, not its bounding rectangle
        else:
            assert n_listed <= (1.6 if kind == "c1" else 6.0) * n_cov + 50, (n_listed, n_cov)  # bounding rectangles, not more
    finally:
        r.close()
        pc.close()
        ctx.close()


def test_stopwatch_stage_and_kernel_times(ws, ctx, oracle):
    """GPUStopwatch (utils.rs:26-134) equivalents: the four stage labels (renderer.rs:221,230, lib.rs:448 + binning)
    and, at level 2, one interval per kernel launch in launch order, plus the empty calibration launch after K1."""
    sc = scenes.c1(ws, oracle, n=5000, viewport=(320, 240), seed=4)
    pc = ws.PointCloud(ctx, sc.gpc)
    r = ws.GaussianRenderer(ctx, "rgba32float", 3, False)
    try:
        r.enable_timers(2)
        r.prepare(pc, sc.args)
        r.render(pc)
        st = r.stage_times()
        assert set(st) >= {"preprocess", "sorting", "binning", "rasterization"}
        assert all(v > 0 for v in st.values())
        kt = r.kernel_times()
        names = [k for k, _ in kt]
        assert names[0] == "k_preprocess" and names[1] == "_empty_launch" and names[-1] == "k_blend"
        assert "k_bin_prefix" in names and "k_bin_emit" in names
        assert sum(n.startswith("depth:") for n in names) == 12          # 4 passes x (histogram, scan, scatter)
        assert all(0 < ms < 50 for _, ms in kt)
        r.enable_timers(0)
        with pytest.raises(ws.WebSplatError):
            r.kernel_times()
    finally:
        r.close()
        pc.close()


def test_wave_stats_capture(ws, ctx, oracle):
    """Analysis read-back of the compositing pass: per wave the staged records it composited, per tile the lock-step
    cost (sum over batches of the busiest wave).  Needs capture mode; the image is unchanged by it."""
    sc = scenes.c1(ws, oracle, n=8000, viewport=(320, 200), seed=5)
    pc = ws.PointCloud(ctx, sc.gpc)
    r = ws.GaussianRenderer(ctx, "rgba32float", 3, False)
    try:
        r.prepare(pc, sc.args)
        r.render(pc)
        img_plain = r.download_target()
        with pytest.raises(ws.WebSplatError):
            r.wave_stats()
        r.enable_capture(True)
        r.prepare(pc, sc.args)
        r.render(pc)
        img_cap = r.download_target()
        # (capture mode always composites whole 32x32 binning tiles; without it a viewport this small is drawn by two
        # 32x16 workgroups per tile, whose tile-local coordinates round the affine map differently in the last bits)
        assert np.abs(img_plain - img_cap).max() <= 2.0 * 2.0 ** -14   # (... and a plain frame may bin at 64x64, see below)
        st = r.wave_stats().astype(np.int64)
        ts = r.tile_stats(with_consumed=True)
        tw, th = ctx.tile_size()
        nw = (tw // 8) * (th // 8)
        assert st.shape == (len(ts["list_len"]), 17)
        per_wave, lock = st[:, :nw], st[:, 16]
        assert np.all(st[:, nw:16] == 0)
        assert np.all(lock >= per_wave.max(axis=1)) and np.all(lock <= per_wave.sum(axis=1))
        assert np.all(per_wave.max(axis=1) <= ts["consumed"] + 3)            # lists are padded to multiples of four
        assert np.all((ts["list_len"] == 0) <= (per_wave.sum(axis=1) == 0))
        assert per_wave.sum() > 0
    finally:
        r.close()
        pc.close()


@pytest.mark.experimental
def test_footprint_modes_draw_the_same_image(ws, oracle, monkeypatch):
    """The footprint word only decides which tiles LIST a splat; the blend's per-pixel test decides what is drawn.  The
    ellipse footprint (WS_FOOTPRINT=ellipse) must therefore give the bit-identical image with fewer tile entries."""
    sc = scenes.c1(ws, oracle, n=20_000, viewport=(800, 600), seed=31)
    out = {}
    monkeypatch.setenv("WS_BIN_SHIFT", "0")  # both at the compositing tile: the entry counts are compared below
    for mode in ("rect", "ellipse"):
        if mode == "ellipse":
            monkeypatch.setenv("WS_FOOTPRINT", "ellipse")
        c = ws.Context(0)
        pc = ws.PointCloud(c, sc.gpc)
        r = ws.GaussianRenderer(c, "rgba32float", 3, False)
        try:
            r.prepare(pc, sc.args)
            r.render(pc, background=(0.1, 0.2, 0.3, 1.0))
            out[mode] = (r.download_target(), r.frame_stats())
        finally:
            r.close()
            pc.close()
            c.close()
    (img_r, st_r), (img_e, st_e) = out["rect"], out["ellipse"]
    assert st_r["overflow"] == 0 and st_e["overflow"] == 0
    assert st_r["num_visible"] == st_e["num_visible"]
    assert st_e["num_tile_entries"] < st_r["num_tile_entries"]
    # (identical while no tile saturates; where one does, the early-out -- checked every four staged records of a wave --
    # falls on another record and the pixels keep or lose contributions below T_MIN = 2^-14: measured 5e-5 on hd1m / c3)
    assert np.abs(img_r.astype(np.float64) - img_e.astype(np.float64)).max() <= 2.0 * 2.0 ** -14


@pytest.mark.parametrize("fmt,mode", [("rgba32float", 0), ("rgba16float", 1), ("rgba8unorm", 2)])
def test_target_precision_blend_equals_the_oracles_per_blend_rounding(ws, ctx, oracle, fmt, mode):
    """WS_BLEND_TARGET_PRECISION: back to front over the clear colour, the destination rounded to the target's precision
    after EVERY splat -- the reference's fixed-function blend on its Rgba16Float / Rgba8Unorm targets (renderer.rs:63-67,
    bin/render.rs:154, bin/measure.rs:184) -- against the oracle's target modes: at most one unit in the last place."""
    sc = scenes.c1(ws, oracle, n=10_000, viewport=(800, 600), seed=41)
    bg = (0.25, 0.5, 0.125, 1.0)
    pc = ws.PointCloud(ctx, sc.gpc)
    r = ws.GaussianRenderer(ctx, fmt, 3, False)
    try:
        r.set_blend_mode("target")
        r.prepare(pc, sc.args)
        r.render(pc, background=bg)
        got = r.download_target()
        fr = r.download_frame()   # the blend alone: the oracle composites the library's own records in the library's order
        ref = oracle.render(fr["splats"], fr["sorted"], 800, 600, bg, mode)
        if mode == 0:
            ok, msg, *_ = scenes.image_close(got, ref, max_abs=2e-5, mean_abs=1e-6, allow_boundary=True)
            assert ok, msg
        elif mode == 1:
            lsb = scenes.half_ulp_diff(got.view(np.uint16), ref.astype(np.float16).view(np.uint16))
            assert (lsb > 1).sum() <= 16 and (lsb > 0).mean() < 1e-3, (int(lsb.max()), float((lsb > 0).mean()))
        else:
            lsb = np.abs(got.astype(np.int64) - np.rint(ref * 255.0).astype(np.int64))
            assert (lsb > 1).sum() <= 16 and (lsb > 0).mean() < 1e-3, (int(lsb.max()), float((lsb > 0).mean()))
        # ... and it is a different image from the fast mode's single rounding wherever many splats overlap
        r.set_blend_mode("fast")
        r.render(pc, background=bg)
        fast = r.download_target()
        if mode == 2:
            assert np.abs(fast.astype(np.int64) - got.astype(np.int64)).max() >= 1
    finally:
        r.close()
        pc.close()


def test_binning_granularity_is_decided_per_frame_on_the_device(ws, oracle, monkeypatch):
    """K1 sums the tiles of every splat's rectangle at the blend's tile size and at twice that size; every later kernel
    derives the same decision from the two sums: bin at 64x64 (four 32x32 compositing workgroups share one list: half the
    entries to emit and sort) when the rectangles shrink by 1.50x or more, else at 32x32.  The image is the same up to the
    early-out granularity, the oracle tolerance holds either way, WS_BIN_SHIFT=0 / 1 force the choice, capture mode always
    sees the blend's own tiles."""
    rng = np.random.default_rng(51)
    big = synth.scene_c1(n=12_000, seed=51)
    ncol = big.shape[1]
    big[:, ncol - 7:ncol - 4] = np.log(rng.uniform(0.03, 0.09, size=(12_000, 3))).astype(np.float32)   # ~40..120 px across
    small = synth.scene_c1(n=40_000, seed=52)
    small[:, ncol - 7:ncol - 4] = np.log(rng.uniform(0.0005, 0.002, size=(40_000, 3))).astype(np.float32)  # about a pixel
    vp = (960, 640)
    cj = synth.camera_c1(*vp)
    cj.fx = cj.fy = 900.0
    out = {}
    for name, rows in (("big", big), ("small", small)):
        sc = scenes.Scene(ws, oracle, rows, 3, cj, vp)
        for mode in ("auto", "0", "1"):
            if mode == "auto":
                monkeypatch.delenv("WS_BIN_SHIFT", raising=False)
            else:
                monkeypatch.setenv("WS_BIN_SHIFT", mode)
            c = ws.Context(0)
            pc = ws.PointCloud(c, sc.gpc)
            r = ws.GaussianRenderer(c, "rgba32float", 3, False)
            try:
                r.prepare(pc, sc.args)
                r.render(pc, background=(0.2, 0.1, 0.3, 1.0))
                img = r.download_target()
                st = r.frame_stats()
                assert st["overflow"] == 0 and r.errors()[0] == 0
                out[name, mode] = (img, st["num_tile_entries"], r.binning_tile())
                if mode == "auto":
                    ref, ofr = sc.oracle_image(pc, background=(0.2, 0.1, 0.3, 1.0))
                    ok, msg, *_ = scenes.image_close(img, ref, proof=sc.proof(ofr, (0.2, 0.1, 0.3, 1.0)))
                    assert ok, (name, msg)
                    r.enable_capture(True)      # parity tooling reads per-tile lists back: always the blend's own tiles
                    r.prepare(pc, sc.args)
                    r.render(pc)
                    assert r.binning_tile() == (32, 32)
            finally:
                r.close()
                pc.close()
                c.close()
    assert out["big", "auto"][2] == (64, 64) and out["small", "auto"][2] == (32, 32)
    assert out["big", "0"][2] == (32, 32) and out["big", "1"][2] == (64, 64) and out["small", "1"][2] == (64, 64)
    assert out["big", "auto"][1] == out["big", "1"][1] < 0.6 * out["big", "0"][1]        # half the entries
    assert out["small", "auto"][1] == out["small", "0"][1]
    assert out["small", "1"][1] > 0.85 * out["small", "0"][1]                            # ... where there is nothing to halve
    for name in ("big", "small"):
        a, b = out[name, "0"][0], out[name, "1"][0]
        assert np.abs(a.astype(np.float64) - b.astype(np.float64)).max() <= 2.0 * 2.0 ** -14
        assert np.array_equal(out[name, "auto"][0], out[name, "1" if name == "big" else "0"][0])


@pytest.mark.experimental
@pytest.mark.parametrize("viewport,n,bins", [((640, 400), 6000, 512), ((1920, 1080), 150_000, 2048), ((352, 288), 6000, 128),
                                             ((256, 192), 6000, 0)])
def test_single_pass_tile_sort_equals_the_digit_passes(ws, oracle, monkeypatch, viewport, n, bins):
    """WS_TILE_SORT=wide: with at most 2048 binning tiles the tile-id sort is ONE stable counting pass over the whole tile id
    (k_bin_emit leaves [sort tile][bin] counts, k_tile_col_scan_wide, k_tile_scatter_wide; the tile ranges are prefix sums
    of the bin totals).  It must produce exactly what the default two digit passes produce: the same ranges, the same
    entries in the same order -- hence the same image, bit for bit.  48 tiles (256x192) stay with the single 6-bit pass."""
    rows = synth.scene_c2(n=n, seed=61) if n > 100_000 else synth.scene_c1(n=n, seed=61)
    cj = (synth.orbit_cameras(8, viewport[0], viewport[1], 1500.0, 1500.0)[3] if n > 100_000 else synth.camera_c1(*viewport))
    if n <= 100_000:
        cj.fx = cj.fy = float(viewport[0])
    out = {}
    for mode in ("wide", "passes"):
        if mode == "wide":
            monkeypatch.setenv("WS_TILE_SORT", "wide")
        else:
            monkeypatch.delenv("WS_TILE_SORT", raising=False)
        c = ws.Context(0)
        gpc = ws.GenericGaussianPointCloud.from_ply_rows(rows, 3)
        cam = ws.PerspectiveCamera.from_scene_camera(cj.position, cj.rotation, cj.fx, cj.fy, *viewport)
        cam.fit_near_far(gpc.aabb)
        args = ws.SplattingArgs(camera=cam, viewport=viewport, max_sh_deg=3)
        pc = ws.PointCloud(c, gpc)
        r = ws.GaussianRenderer(c, "rgba32float", 3, False)
        try:
            r.enable_timers(2)
            for _ in range(2):  # twice: the scratch (count rows, totals) is reused without being cleared
                r.prepare(pc, args)
                r.render(pc, background=(0.1, 0.2, 0.3, 1.0))
            labels = [k for k, _ in r.kernel_times()]
            st = r.frame_stats()
            assert st["overflow"] == 0 and r.errors()[0] == 0 and st["num_tile_entries"] > 2048 * 2
            out[mode] = (r.download_target(), r.tile_lists(), st, labels.count("tiles:k_sort_scatter"), r.binning_tile())
        finally:
            r.close()
            pc.close()
            c.close()
    (img_w, (b_w, e_w, l_w), st_w, passes_w, bt_w), (img_p, (b_p, e_p, l_p), st_p, passes_p, bt_p) = out["wide"], out["passes"]
    assert passes_w == 1 and passes_p == (2 if bins else 1)
    assert bt_w == bt_p and st_w == st_p
    assert np.array_equal(b_w, b_p) and np.array_equal(e_w, e_p)
    assert np.array_equal(l_w, l_p)
    assert np.array_equal(img_w, img_p)


def test_compositing_workgroups_run_longest_list_first(ws, oracle, monkeypatch):
    """k_blend_order (round 5): behind the tile-id sort one workgroup orders the blend's tiles by the length of their lists,
    longest first, and hands every compositing workgroup its tile AND that tile's entry range.  The table must be a
    permutation of the frame's tiles with the ranges of THIS frame (two different views in a row on one renderer: a stale
    table or stale ranges would belong to the other view), non-increasing in length class; and the image must be the image
    of the same frame composited in image order (WS_BLEND_ORDER=0), bit for bit, for both binning granularities."""
    rng = np.random.default_rng(61)
    rows = synth.scene_c1(n=60_000, seed=61)
    ncol = rows.shape[1]
    rows[:, ncol - 7:ncol - 4] = np.log(rng.uniform(0.004, 0.05, size=(60_000, 3))).astype(np.float32)
    # 20 x 18 tiles = 90 blocks of 2 x 2 tiles: the image-order grid is padded to 96 blocks, and its six invalid workgroup
    # indices (354 .. 359) are BELOW the tile count -- an ordered launch must not apply the image-order layout's validity test
    vp = (640, 576)
    cams = synth.orbit_cameras(5, vp[0], vp[1], 600.0, 600.0, radius=3.0, height_off=0.4)
    imgs = {}
    for order in ("1", "0"):
        for shift in ("auto", "0"):
            monkeypatch.setenv("WS_BLEND_ORDER", order)
            if shift == "auto":
                monkeypatch.delenv("WS_BIN_SHIFT", raising=False)
            else:
                monkeypatch.setenv("WS_BIN_SHIFT", shift)
            c = ws.Context(0)
            sc = scenes.Scene(ws, oracle, rows, 3, cams[0], vp)
            pc = ws.PointCloud(c, sc.gpc)
            r = ws.GaussianRenderer(c, "rgba32float", 3, False)
            try:
                for k, cj in enumerate((cams[0], cams[2], cams[3], cams[2])):
                    cam = ws.PerspectiveCamera.from_scene_camera(cj.position, cj.rotation, cj.fx, cj.fy, *vp)
                    cam.fit_near_far(sc.gpc.aabb)
                    args = ws.SplattingArgs(camera=cam, viewport=vp, max_sh_deg=3)
                    r.prepare(pc, args)
                    r.render(pc, background=(0.1, 0.2, 0.3, 1.0))
                    imgs[order, shift, k] = r.download_target().copy()
                    assert r.frame_stats()["overflow"] == 0 and r.errors()[0] == 0
                    tab = r.blend_order()
                    if order == "0":
                        assert tab.shape[0] == 0
                        continue
                    tw, th = c.tile_size()
                    tiles_x, tiles_y = -(-vp[0] // tw), -(-vp[1] // th)
                    nt = tiles_x * tiles_y
                    assert tab.shape[0] >= nt and (tab[nt:, 0] == 0xFFFFFFFF).all()
                    code = tab[:nt, 0].astype(np.int64)
                    tile = (code >> 16) * tiles_x + (code & 0xFFFF)
                    assert sorted(tile.tolist()) == list(range(nt))                      # every tile exactly once
                    begin, end, _ = r.tile_lists()                                       # per LIST (binning tile)
                    bw, _ = r.binning_tile()
                    s = 1 if bw > tw else 0
                    lx = (tiles_x + s) >> s
                    lidx = ((code >> 16) >> s) * lx + ((code & 0xFFFF) >> s)
                    want_b = np.where(end[lidx] > begin[lidx], begin[lidx], 0)
                    want_e = np.where(end[lidx] > begin[lidx], end[lidx], 0)
                    assert np.array_equal(tab[:nt, 1], want_b) and np.array_equal(tab[:nt, 2], want_e)   # THIS frame's ranges
                    cls = np.minimum((tab[:nt, 2] - tab[:nt, 1]) >> 4, 2047)
                    assert (np.diff(cls.astype(np.int64)) <= 0).all()                    # longest first
                    assert cls[0] > cls[-1]
            finally:
                r.close()
                pc.close()
                c.close()
    for shift in ("auto", "0"):
        for k in range(4):
            assert np.array_equal(imgs["1", shift, k], imgs["0", shift, k]), (shift, k)
    assert np.array_equal(imgs["1", "auto", 1], imgs["1", "auto", 3])            # the same view again: the same image
    assert not np.array_equal(imgs["1", "auto", 0], imgs["1", "auto", 1])


def _row_at(row, xyz):
    r = np.array(row, copy=True)[None, :]
    r[0, 0:3] = np.asarray(xyz, dtype=np.float32)
    return r


@pytest.mark.parametrize("digit_bits", [8, 9])
def test_depth_sort_skips_its_last_pass_on_a_narrow_key_range(ws, oracle, monkeypatch, digit_bits):
    """Round 5: a camera outside the scene sees depth keys -- bits(zfar - z), preprocess.wgsl:270-273 -- that span less than 2^24;
    sorted as (key - base) the fourth 8-bit pass of the LSD sort is then the identity, and the frame's depth sort executes three
    passes (decided on the device from the key range K1 stored; the readers of the sorted arrays follow).  Draw order, tile
    lists and image must be exactly those of the same frame with the pass forced (WS_DEPTH_SKIP_TOP=0) -- and of the oracle's
    stable sort; the same slab with a trail of splats far behind it (keys over a factor of > 4: more than 2^24 apart) keeps
    its four passes.  Round 6, digit_bits = 9: three 9-bit passes cover key - base below 2^27 (both frames here), a fourth over
    bits 27..31 runs only beyond that (a frame with a splat right at the camera and one at the far plane: keys 2^30 apart)."""
    monkeypatch.setenv("WS_DEPTH_DIGIT_BITS", str(digit_bits))
    span = 1 << (3 * digit_bits)
    low = (1 << digit_bits) - 1
    vp = (800, 600)
    rows = synth.scene_c1(n=120_000, seed=71)
    rows[:, 2] *= 0.5        # a slab: seen from outside its keys span a factor of ~2.5 (a factor of 4 is 2^24 in f32 bits)
    outside = synth.look_at_camera(0, [0.2, -0.3, -9.0], [0, 0, 0], vp[0], vp[1], 2100.0, 2100.0)
    trail = synth.scene_c1(n=4000, seed=72)
    trail[:, 0:2] *= 0.3
    trail[:, 2] = np.random.default_rng(73).uniform(2.0, 60.0 if digit_bits == 8 else 3000.0, size=4000).astype(np.float32)   # far behind the slab, along the view axis
    if digit_bits == 9:
        trail[:8, 2] = -8.9   # ... and a few right in front of the camera: bits(zfar - z) of near and far splats are > 2^27 apart
    rows_wide = np.concatenate([rows, trail])
    got = {}
    for skip in ("1", "0"):
        monkeypatch.setenv("WS_DEPTH_SKIP_TOP", skip)
        c = ws.Context(0)
        try:
            for name, cj, rr in (("outside", outside, rows), ("inside", outside, rows_wide)):
                sc = scenes.Scene(ws, oracle, rr, 3, cj, vp)
                if digit_bits == 9 and name == "inside":
                    # 2^27 in f32 bits is a factor of 65 536 in (zfar - z), and K1 culls z_ndc >= 1 (preprocess.wgsl:190): a splat can
                    # come that close to the far plane only when znear / zfar is large (1 - z_ndc ~ (znear / zfar) * eps must stay
                    # above an f32 ulp) -- no camera fitted to a scene gets there, which is why three 9-bit passes are the rule.
                    # Forced here: row 0 ON the view axis 3000 units out, the far plane 1e-5 (relative) behind it, the near plane at 300.
                    fwd = np.asarray(cj.rotation, dtype=np.float64)[:, 2]
                    far = np.asarray(cj.position, dtype=np.float64) + 3000.0 * fwd
                    sc = scenes.Scene(ws, oracle, np.concatenate([_row_at(rr[0], far), rr[1:]]), 3, cj, vp)
                    sc.args.camera.znear = 300.0
                    sc.args.camera.zfar = float(np.float32(3000.0 * (1.0 + 1.0e-5)))
                pc = ws.PointCloud(c, sc.gpc)
                r = ws.GaussianRenderer(c, "rgba32float", 3, False)
                try:
                    for _ in range(2):                       # (twice: the second frame reuses every buffer of the first)
                        r.prepare(pc, sc.args)
                        r.render(pc)
                    fr = r.download_frame()
                    b, e, ent = r.tile_lists()
                    got[name, skip] = (r.depth_sort_passes(), fr["keys"], fr["sorted"], b, e, ent, r.download_target())
                    assert r.frame_stats()["overflow"] == 0 and r.errors()[0] == 0
                    if skip == "1":
                        _, order = oracle.sort_pairs(fr["keys"], np.arange(fr["num_visible"], dtype=np.uint32))
                        assert np.array_equal(fr["sorted"], order), name                 # the stable sort's permutation
                        ref, ofr = sc.oracle_image(pc)
                        ok, msg, *_ = scenes.image_close(got[name, skip][6], ref, proof=sc.proof(ofr))
                        assert ok, (name, msg)
                finally:
                    r.close()
                    pc.close()
        finally:
            c.close()
    k_out, k_in = got["outside", "1"][1].astype(np.int64), got["inside", "1"][1].astype(np.int64)
    assert k_out.max() - (k_out.min() & ~low) < span <= k_in.max() - (k_in.min() & ~low), (k_out.min(), k_out.max(), k_in.min(), k_in.max())
    assert got["outside", "1"][0] == 3 and got["outside", "0"][0] == 4
    assert got["inside", "1"][0] == 4 and got["inside", "0"][0] == 4
    for name in ("outside", "inside"):
        for k in range(1, 7):
            assert np.array_equal(got[name, "1"][k], got[name, "0"][k]), (name, k)


def test_depth_sort_digit_width_follows_the_previous_frames_key_range(ws, oracle, monkeypatch):
    """Round 6: with no width forced, a renderer sorts a frame with 8-bit digits when its previous frame's keys spanned < 2^24
    (three passes), with 9-bit digits when they did not (three passes again, instead of four 8-bit ones) -- the answer travels
    through pinned memory, no sync.  The first frame knows nothing and takes 8 bits; the order -- hence the image -- is the
    stable sort's whichever width ran."""
    monkeypatch.delenv("WS_DEPTH_DIGIT_BITS", raising=False)
    vp = (800, 600)
    rows = synth.scene_c1(n=60_000, seed=75)
    rows[:, 2] *= 0.5
    cam = synth.look_at_camera(0, [0.2, -0.3, -9.0], [0, 0, 0], vp[0], vp[1], 2100.0, 2100.0)
    trail = synth.scene_c1(n=3000, seed=76)
    trail[:, 0:2] *= 0.3
    trail[:, 2] = np.random.default_rng(77).uniform(2.0, 60.0, size=3000).astype(np.float32)
    c = ws.Context(0)
    try:
        for rr, want_bits, want_passes in ((rows, [8, 8, 8], [3, 3, 3]), (np.concatenate([rows, trail]), [8, 9, 9], [4, 3, 3])):
            sc = scenes.Scene(ws, oracle, rr, 3, cam, vp)
            pc = ws.PointCloud(c, sc.gpc)
            r = ws.GaussianRenderer(c, "rgba32float", 3, False)
            try:
                imgs, bits, passes = [], [], []
                for _ in range(3):
                    r.prepare(pc, sc.args)
                    r.render(pc)
                    c.sync()                       # (the mailbox word of this frame is posted: the next prepare() sees it)
                    bits.append(r.depth_sort_digit_bits())
                    passes.append(r.depth_sort_passes())
                    imgs.append(r.download_target())
                    fr = r.download_frame()
                    _, order = oracle.sort_pairs(fr["keys"], np.arange(fr["num_visible"], dtype=np.uint32))
                    assert np.array_equal(fr["sorted"], order)
                assert bits == want_bits and passes == want_passes, (bits, passes)
                assert np.array_equal(imgs[0], imgs[1]) and np.array_equal(imgs[0], imgs[2])
            finally:
                r.close()
                pc.close()
    finally:
        c.close()


def test_workgroup_order_follows_how_the_context_is_driven(ws, oracle, monkeypatch):
    """The automatic choice of the blend's workgroup order (round 5): a renderer that draws one frame at a time orders its tiles
    longest list first (no tail of idle slots: blend -12 ... -18 %); renderers that draw in turn on a ring of streams -- frames
    in flight, whether through ws_view_batch or through the caller's own loop -- keep the image order (longest-first measured
    -9 ... -17 % frames/s there).  The library tells the two apart by whether consecutive prepare() calls of the context stay on
    one stream; the image is the same either way."""
    import ctypes as C
    monkeypatch.delenv("WS_BLEND_ORDER", raising=False)
    vp = (640, 576)
    rng = np.random.default_rng(81)
    rows = synth.scene_c1(n=40_000, seed=81)
    ncol = rows.shape[1]
    rows[:, ncol - 7:ncol - 4] = np.log(rng.uniform(0.004, 0.05, size=(40_000, 3))).astype(np.float32)
    cams = synth.orbit_cameras(4, vp[0], vp[1], 600.0, 600.0, radius=3.0, height_off=0.4)
    c = ws.Context(0)
    sc = scenes.Scene(ws, oracle, rows, 3, cams[0], vp)
    pc = ws.PointCloud(c, sc.gpc)
    hip = C.CDLL("libamdhip64.so")
    streams = [C.c_void_p() for _ in range(2)]
    for s in streams:
        assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0  # hipStreamNonBlocking
    rs = [ws.GaussianRenderer(c, "rgba32float", 3, False) for _ in range(2)]
    batch = ws.ViewBatch(c, "rgba32float", 3, False, 2)
    try:
        # two renderers in turn on two streams: frames in flight -> image order (no table), from the first frame on
        for i in range(6):
            k = i % 2
            rs[k].prepare(pc, sc.args, stream=streams[k].value)
            rs[k].render(pc, stream=streams[k].value)
        for k in range(2):
            c.sync(streams[k].value)
            assert rs[k].blend_order().shape[0] == 0
        in_flight = rs[0].download_target().copy()
        # one of them alone, frame after frame on its stream: after four calls it orders its tiles
        for i in range(6):
            rs[0].prepare(pc, sc.args, stream=streams[0].value)
            rs[0].render(pc, stream=streams[0].value)
        c.sync(streams[0].value)
        assert rs[0].blend_order().shape[0] > 0
        assert np.array_equal(rs[0].download_target(), in_flight)
        # the slots of a view batch with two frames in flight never order, however many frames they draw
        targets = [c.malloc(vp[0] * vp[1] * 16) for _ in range(2)]
        try:
            batch.render(pc, [sc.args] * 12, [targets[i % 2] for i in range(12)], vp[0] * 16)
            batch.sync()
            assert batch.errors() == 0
            for slot in range(2):
                assert batch.renderer(slot).blend_order().shape[0] == 0
            got = c.download(targets[1], (vp[1], vp[0], 4), np.float32)
            assert np.array_equal(got, in_flight)
        finally:
            for t in targets:
                c.free(t)
    finally:
        batch.close()
        for r in rs:
            r.close()
        for s in streams:
            hip.hipStreamDestroy(s)
        pc.close()
        c.close()


def test_two_contexts_in_one_process_draw_the_same_frames(ws, oracle):
    """Round 6 (verdict r05 item 8b): a process may hold several ws_contexts (several scenes, several tenants of one device; on a
    multi-GPU host one per device ordinal).  Two contexts on the device, each with its own point cloud, renderer and stream, drawn in
    turn: every frame equals the one a lone context draws, no error bit, and destroying one context leaves the other working."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    sc = scenes.c1(ws, oracle, n=30_000, viewport=(640, 480), seed=81)
    cams = synth.orbit_cameras(6, 640, 480, 640.0, 640.0)
    views = []
    for cj in cams:
        cam = ws.PerspectiveCamera.from_scene_camera(cj.position, cj.rotation, cj.fx, cj.fy, 640, 480)
        cam.fit_near_far(sc.gpc.aabb)
        views.append(ws.SplattingArgs(camera=cam, viewport=(640, 480), max_sh_deg=3))
    ref = []
    c0 = ws.Context(0)
    pc0 = ws.PointCloud(c0, sc.gpc)
    r0 = ws.GaussianRenderer(c0, "rgba32float", 3, False)
    for v in views:
        r0.prepare(pc0, v)
        r0.render(pc0)
        ref.append(r0.download_target())
    a, b = c0, ws.Context(0)
    assert a.handle.value != b.handle.value
    streams = []
    for _ in range(2):
        s = C.c_void_p()
        assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0
        streams.append(s)
    pcb = ws.PointCloud(b, sc.gpc)
    rb = ws.GaussianRenderer(b, "rgba32float", 3, False)
    try:
        for i, v in enumerate(views):          # interleaved: context a draws view i while context b draws view (i + 3) % 6
            w = views[(i + 3) % 6]
            r0.prepare(pc0, v, stream=streams[0].value)
            rb.prepare(pcb, w, stream=streams[1].value)
            r0.render(pc0, stream=streams[0].value)
            rb.render(pcb, stream=streams[1].value)
            a.sync(streams[0].value)
            b.sync(streams[1].value)
            assert np.array_equal(r0.download_target(), ref[i]), i
            assert np.array_equal(rb.download_target(), ref[(i + 3) % 6]), i
            assert r0.errors()[0] == 0 and rb.errors()[0] == 0
        rb.close()
        pcb.close()
        b.close()                               # one context goes away ...
        r0.prepare(pc0, views[2])
        r0.render(pc0)
        assert np.array_equal(r0.download_target(), ref[2])   # ... the other is unharmed
    finally:
        r0.close()
        pc0.close()
        a.close()
        for s in streams:
            hip.hipStreamDestroy(s)


@pytest.mark.experimental
@pytest.mark.parametrize("fmt", ["rgba32float", "rgba16float", "rgba8unorm"])
@pytest.mark.parametrize("order", ["0", "1"])
def test_async_blend_is_bit_identical(ws, oracle, monkeypatch, fmt, order):
    """Round 6: k_blend2 (double-buffered staging, LDS arrival counters, no per-batch workgroup barrier) composites every pixel
    with the statements of k_blend in the same order -- only WHEN a wave does its share differs.  Its image must therefore be
    the bit-identical image: tiles with one staged batch, tiles with dozens (400 k splats on 320x240: lists of > 10 k entries,
    early saturation in the dense centre), empty tiles, tiles cut by the image edge (330x250), both workgroup orders."""
    monkeypatch.setenv("WS_BLEND_ORDER", order)
    cases = [("sparse", synth.scene_c1(n=20_000, seed=91), synth.camera_c1(330, 250), (330, 250)),
             ("dense", synth.scene_c2(n=400_000, seed=92), synth.orbit_cameras(8, 320, 240, 320.0, 320.0)[2], (320, 240)),
             ("hd", synth.scene_c2(n=300_000, seed=93), synth.orbit_cameras(8, 1280, 720, 1280.0, 1280.0)[5], (1280, 720))]
    imgs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("WS_BLEND_ASYNC", mode)
        c = ws.Context(0)
        try:
            for name, rows, cj, vp in cases:
                sc = scenes.Scene(ws, oracle, rows, 3, cj, vp)
                pc = ws.PointCloud(c, sc.gpc)
                r = ws.GaussianRenderer(c, fmt, 3, False)
                try:
                    for rep in range(2):
                        r.prepare(pc, sc.args)
                        r.render(pc, background=(0.1, 0.2, 0.3, 0.4))
                        imgs[mode, name, rep] = r.download_target()
                    st = r.frame_stats()
                    assert st["overflow"] == 0 and r.errors()[0] == 0, (mode, name, st)
                    if name == "dense":
                        b, e, _ = r.tile_lists()
                        assert (e - b).max() > 4 * 512        # lists of many staged batches
                finally:
                    r.close()
                    pc.close()
        finally:
            c.close()
    for name, *_ in cases:
        for rep in range(2):
            assert np.array_equal(imgs["0", name, rep], imgs["1", name, rep]), (name, rep)
        assert np.array_equal(imgs["1", name, 0], imgs["1", name, 1]), name


def test_frame_trace_stamps_k1_and_the_blend_on_the_device_clock(ws, ctx, oracle):
    """Round 6: ws_renderer_enable_frame_trace -- K1 and the compositing kernel of the next n frames leave {first workgroup start,
    last workgroup end} on the 100-MHz device clock (the instrument of scripts/inflight_device_trace.py: rocprofv3 serialises
    frames in flight).  Per frame K1 starts before it ends, ends before the blend starts (the sorts and the binning lie between),
    frames of one renderer follow one another; the image is the untraced image; frames beyond n are not traced."""
    sc = scenes.c1(ws, oracle, n=30_000, viewport=(640, 480), seed=95)
    pc = ws.PointCloud(ctx, sc.gpc)
    r = ws.GaussianRenderer(ctx, "rgba32float", 3, False)
    try:
        r.prepare(pc, sc.args)
        r.render(pc)
        ref = r.download_target()
        assert r.frame_trace().shape == (0, 4)
        r.enable_frame_trace(3)
        for _ in range(5):
            r.prepare(pc, sc.args)
            r.render(pc)
        tr = r.frame_trace().astype(np.int64)
        assert tr.shape == (3, 4)
        assert np.array_equal(r.download_target(), ref)
        for k1s, k1e, bs, be in tr:
            assert 0 < k1s < k1e <= bs < be, (k1s, k1e, bs, be)
            assert (k1e - k1s) < 100 * 1000 and (be - bs) < 100 * 1000      # (100-MHz ticks: well under a millisecond each)
        assert np.all(tr[1:, 0] >= tr[:-1, 3])                                # one renderer, one stream: frame after frame
        r.enable_frame_trace(0)
        r.prepare(pc, sc.args)
        r.render(pc)
        assert r.frame_trace().shape == (0, 4) and np.array_equal(r.download_target(), ref)
    finally:
        r.close()
        pc.close()


@pytest.mark.parametrize("seed", [101, 102, 103, 104, 105, 106])
def test_exact_cut_mode_needs_no_boundary_allowance(ws, ctx, oracle, seed):
    """Round 6: in WS_BLEND_FAST_EXACT_CUT mode the keep / discard decision of a fragment at the cut-off (gaussian.wgsl:61) is the
    reference's own expression, so the f32 image equals the oracle's within the plain tolerance on EVERY pixel -- no cut-off boundary
    allowance, no proof -- over seeds, viewports that cut tiles, opaque and transparent clear colours; and the fast mode differs from
    it only in a handful of pixels, each by less than one boundary fragment's weight."""
    rng = np.random.default_rng(seed)
    vp = [(640, 480), (801, 599), (330, 250), (1024, 768), (512, 512), (1280, 720)][seed - 101]
    n = int(rng.integers(8_000, 60_000))
    sc = scenes.c1(ws, oracle, n=n, viewport=vp, seed=seed)
    bg = (0.0, 0.0, 0.0, 0.0) if seed % 2 else (0.2, 0.1, 0.3, 1.0)
    pc = ws.PointCloud(ctx, sc.gpc)
    r = ws.GaussianRenderer(ctx, "rgba32float", 3, False)
    try:
        r.prepare(pc, sc.args)
        r.render(pc, background=bg)
        fast = r.download_target()
        r.set_blend_mode("fast_exact_cut")
        r.render(pc, background=bg)
        exact = r.download_target()
        r.set_blend_mode("fast")
        assert r.errors()[0] == 0
        ref, _ = sc.oracle_image(pc, background=bg)
        ok, msg, mx, mean, nb = scenes.image_close(exact, ref, allow_boundary=False)
        assert ok and nb == 0, (msg, mx)
        assert mx <= 5e-4, mx                                        # (the early-out bound is 6.1e-5 x the colour range)
        d = np.abs(fast.astype(np.float64) - exact.astype(np.float64)).max(axis=2)
        assert (d > 0).sum() <= max(8, 4e-5 * d.size) and d.max() <= scenes.BOUNDARY_STEP, ((d > 0).sum(), d.max())
    finally:
        r.close()
        pc.close()
