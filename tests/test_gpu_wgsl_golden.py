"""The HIP kernels against the reference's OWN shader source (GPU, through the C ABI).

tests/golden/wgsl_*.npz: outputs of preprocess.wgsl / preprocess_compressed.wgsl / gaussian.wgsl executed from their source
text on seeded inputs (generator and interpreter: tests/golden/gen_wgsl_golden.py, oracle/wgsl_exec.py; CPU side of the
same vectors: tests/test_wgsl_golden.py).  Nothing here reads the reference checkout or calls the oracle: the fixtures
are the checker.

Tolerances are those of tests/test_gpu_preprocess.py (K1 / K1c: equal visible set and store order, every f16 field within
one ulp, depth keys within 2 ulp) and, for single fragments, 2e-6 + 1e-4 relative on the premultiplied output."""
import ctypes as C
import os

import numpy as np
import pytest

import scenes
import wgsl_cases
from test_gpu_preprocess import _compare, _prepare

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, "wgsl_%s.npz" % name))


def _bytes(s):
    return bytes(C.string_at(C.byref(s), C.sizeof(s)))


@pytest.mark.parametrize("case", wgsl_cases.K1_CASES)
def test_k1_equals_the_reference_shader(ws, ctx, oracle, case):
    z = load("k1_" + case)
    sc = wgsl_cases.k1_scene(ws, oracle, case)   # (`oracle` only parameterises the Scene helper; it is not called)
    assert np.array_equal(np.ascontiguousarray(sc.gpc.gaussians).view(np.uint8).reshape(-1), z["gaussians"].reshape(-1))
    pc, frame, stats = _prepare(ws, ctx, sc)
    try:
        # the uniform bytes the shader decoded with WGSL's own layout rules are the bytes the library's kernels read
        assert _bytes(sc.args.camera.uniform(sc.viewport)) == z["camera_uniform"].tobytes()
        assert _bytes(pc.settings_uniform(sc.args)) == z["settings_uniform"].tobytes()
        o = z["splats"].copy()
        oh = o.view(np.uint16).reshape(-1, 10)
        undefined = ((oh[:, :4] & 0x7FFF) > 0x7C00).any(axis=1)   # normalize((0,0)): indeterminate in WGSL
        assert undefined.any() == (case in ("fade_in", "extremes", "kernel_0"))
        if undefined.any():
            assert frame["num_visible"] == len(z["keys"])
            gh = frame["splats"].view(np.uint16).reshape(-1, 10)
            ax = gh[undefined][:, :4].view(np.float16).astype(np.float32)
            assert np.isfinite(ax).all() and (ax[:, 1] == 0).all() and (ax[:, 2] == 0).all() and (ax[:, 0] >= 0).all()  # (kernel_0: a zero covariance has lambda1 = 0)
            oh[undefined, :4] = gh[undefined, :4]
        inexact = _compare(frame, o, z["keys"], z["src_index"])
        assert stats["overflow"] == 0 and inexact <= 0.02
    finally:
        pc.close()


@pytest.mark.parametrize("case", wgsl_cases.K1C_CASES)
def test_k1c_matches_the_reference_shader(ws, ctx, case):
    z = load("k1c_" + case)
    gpc, cam, viewport, sh_deg = wgsl_cases.k1c_inputs(ws, case)
    assert np.array_equal(np.ascontiguousarray(gpc.gaussians).view(np.uint8).reshape(-1), z["gaussians"].reshape(-1))
    assert np.array_equal(np.ascontiguousarray(gpc.covars).view(np.uint8).reshape(-1), z["covars"].reshape(-1))
    args = ws.SplattingArgs(camera=cam, viewport=viewport, max_sh_deg=sh_deg)
    pc = ws.PointCloud(ctx, gpc)
    r = ws.GaussianRenderer(ctx, "rgba32float", sh_deg, True)
    try:
        assert _bytes(cam.uniform(viewport)) == z["camera_uniform"].tobytes()
        assert _bytes(pc.settings_uniform(args)) == z["settings_uniform"].tobytes()
        r.enable_capture(True)
        r.prepare(pc, args)
        frame = r.download_frame(with_src_index=True)
        # exp(scaling factor): ocml on the GPU, numpy in the fixture -> axes compared through the covariance they encode
        _compare(frame, z["splats"], z["keys"], z["src_index"], max_inexact_frac=0.05, axes_by_cov=viewport)
    finally:
        r.close()
        pc.close()


@pytest.mark.parametrize("fixture,source,scene", [("k6_fragments", "k1_default", "default"),
                                                  ("k6_fragments_opaque", "frame_opaque", "frame_opaque")])
def test_fragments_equal_the_reference_shader(ws, ctx, oracle, fixture, source, scene):
    """gaussian.wgsl vs_main + fs_main from source, for 48 splats at ~4000 pixel centres, against k_blend drawing the same
    Gaussian ALONE (a one-point cloud with the scene's bounding box, centre and camera, so K1 emits the same Splat record):
    after one splat on a transparent target a pixel holds exactly the fragment's premultiplied output.
    `k6_fragments_opaque`: alpha = 1.0 splats sampled around their centres -- the fragments that reach `min(0.99, .)`."""
    z = load(fixture)
    k1 = load(source)
    sc = wgsl_cases.k1_scene(ws, oracle, scene)
    w, h = sc.viewport
    keep = z["frag_keep"].astype(bool)
    a = (z["frag_screen_pos"].astype(np.float64) ** 2).sum(axis=1)
    near_cut = np.abs(a - scenes.CUT_A) < 1e-4
    r = ws.GaussianRenderer(ctx, "rgba32float", 3, False)
    r.enable_capture(True)
    checked = worst = 0
    try:
        for s in z["picked"]:
            sel = z["frag_splat"] == s
            if not sel.any():
                continue
            i = int(k1["src_index"][s])
            one = ws.GenericGaussianPointCloud(sc.gpc.gaussians[i:i + 1].copy(), sc.gpc.sh_coefs[i:i + 1].copy(), 3, 1,
                                               sc.gpc.aabb, sc.gpc.center)
            pc = ws.PointCloud(ctx, one)
            try:
                r.prepare(pc, sc.args)
                frame = r.download_frame(with_src_index=True)
                assert frame["num_visible"] == 1
                d = scenes.half_ulp_diff(frame["splats"].view(np.uint16).reshape(-1, 10), k1["splats"][s].view(np.uint16).reshape(1, 10))
                assert d.max() <= 1
                r.render(pc)
                img = r.download_target()
            finally:
                pc.close()
            if d.max() > 0:
                continue   # (a K1 field one ulp off moves the whole footprint: K1's own tolerance, tested above)
            px = z["frag_pixel"][sel]
            got = img[px[:, 1], px[:, 0]]
            want = z["frag_out"][sel]
            k, nc = keep[sel], near_cut[sel]
            drawn = got[:, 3] > 0
            assert np.array_equal(drawn[~nc], k[~nc]), "kept / discarded set differs away from the cut-off"
            both = drawn & k
            err = np.abs(got[both] - want[both]) - 1e-4 * np.abs(want[both])
            worst = max(worst, float(err.max()) if both.any() else 0.0)
            assert (err <= 2e-6).all(), float(err.max())
            checked += int(both.sum())
    finally:
        r.close()
    assert checked > (800 if fixture == "k6_fragments" else 300), checked
    if fixture == "k6_fragments_opaque":
        assert (z["frag_out"][keep, 3] == np.float32(0.99)).sum() >= 40


@pytest.mark.parametrize("case", wgsl_cases.FRAME_CASES)
def test_frame_equals_the_reference_shaders(ws, ctx, oracle, case):
    """The whole HIP frame (K1 -> depth sort -> binning -> tile sort -> blend) against the frame the reference's shaders
    draw when executed from source (tests/golden/wgsl_frame*.npz; oblique camera; transparent and opaque clear colour):
    the stated image tolerance, boundary pixels proven."""
    z = load(case)
    sc = wgsl_cases.k1_scene(ws, oracle, case)
    background = tuple(float(x) for x in z["background"])
    assert background == wgsl_cases.FRAME_BACKGROUND[case]
    w, h = sc.viewport
    pc = ws.PointCloud(ctx, sc.gpc)
    r = ws.GaussianRenderer(ctx, "rgba32float", 3, False)
    try:
        assert _bytes(sc.args.camera.uniform(sc.viewport)) == z["camera_uniform"].tobytes()
        assert _bytes(pc.settings_uniform(sc.args)) == z["settings_uniform"].tobytes()
        r.prepare(pc, sc.args)
        r.render(pc, background=background)
        img = r.download_target()
        assert r.frame_stats()["num_visible"] == int(z["num_visible"])
        proof = lambda: scenes.BoundaryProof(z["splats"], z["order"], w, h, background)  # noqa: E731
        ok, msg, mx, mean, nb = scenes.image_close(img, z["image"], proof=proof)
        assert ok, msg
        assert mean < 2e-5, mean
    finally:
        r.close()
        pc.close()


@pytest.mark.parametrize("case", ["small", "two_blocks"])
def test_sorters_equal_the_reference_shader(ws, ctx, case):
    """GPURSSorter (generic path and the depth-sort specialisation, with a companion value) against radix_sort.wgsl
    executed from source (tests/golden/wgsl_sort_*.npz): identical keys and payload, ties in input order."""
    z = load("sort_" + case)
    k, p = z["keys_in"], z["payload_in"]
    sorter = ws.GPURSSorter(ctx, len(k))
    try:
        gk, gp = sorter.sort_host(k, p)
        assert np.array_equal(gk, z["keys_out"]) and np.array_equal(gp, z["payload_out"])
        aux = p * np.uint32(7) + np.uint32(3)
        dk, dp, da = sorter.sort_host(k, p, depth=True, aux=aux)
        assert np.array_equal(dk, z["keys_out"]) and np.array_equal(dp, z["payload_out"]) and np.array_equal(da, aux[z["payload_out"]])
    finally:
        sorter.close()
