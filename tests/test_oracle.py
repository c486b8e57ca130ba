"""CPU tests of the oracle itself: pin it on everything the reference pins (the test_sort vector), on
third-party contracts it restates (IEEE binary16 = numpy float16), and on closed forms of the reference's
math (camera.rs, gaussian.wgsl)."""
import numpy as np
import pytest

from websplat import synth


# ---- half 2.6.0 f16::from_f32 / to_f32 == IEEE binary16 RTE == numpy float16 -------------------------
def test_f16_decode_all_65536(oracle):
    bits = np.arange(65536, dtype=np.uint16)
    want = bits.view(np.float16).astype(np.float32)
    got = np.array([oracle.f16_to_f32(int(b)) for b in bits], dtype=np.float32)
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got[~nan].view(np.uint32), want[~nan].view(np.uint32))


def test_f16_encode_rte(oracle):
    rng = np.random.default_rng(0)
    vals = np.concatenate([
        rng.standard_normal(20000).astype(np.float32) * 10.0 ** rng.uniform(-9, 6, 20000).astype(np.float32),
        # exact ties between neighbouring halves, subnormal range, overflow boundary
        np.array([1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11, 65504.0, 65519.99, 65520.0, 1e9, 2.0 ** -24, 2.0 ** -25,
                  2.0 ** -25 * 1.0001, 6.0e-8, 5.96e-8, 0.0, -0.0, np.inf, -np.inf], dtype=np.float32)])
    with np.errstate(over="ignore"):
        want = vals.astype(np.float16).view(np.uint16)
    got = np.array([oracle.f32_to_f16(float(v)) for v in vals], dtype=np.uint16)
    assert np.array_equal(got, want)


# ---- the reference's only known-answer vector: GPURSSorter::test_sort (gpu_rs.rs:295-331) ------------
def test_sort_known_answer_oracle(oracle):
    n = 8192
    scrambled = np.arange(n - 1, -1, -1, dtype=np.float32)
    keys, payload = oracle.sort_pairs(scrambled.view(np.uint32), np.arange(n, dtype=np.uint32))
    assert np.array_equal(keys.view(np.float32), np.arange(n, dtype=np.float32))
    assert np.array_equal(payload, np.arange(n - 1, -1, -1, dtype=np.uint32))


@pytest.mark.parametrize("n,distinct", [(1, 1), (255, 3), (3840, 1), (3841, 7), (100_000, 1 << 32), (65536, 16)])
def test_sort_stable_oracle(oracle, n, distinct):
    rng = np.random.default_rng(n)
    keys = rng.integers(0, distinct, size=n, dtype=np.uint64).astype(np.uint32)
    k, p = oracle.sort_pairs(keys, np.arange(n, dtype=np.uint32))
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(p, order.astype(np.uint32))
    assert np.array_equal(k, keys[order])


# ---- camera.rs closed forms ---------------------------------------------------------------------------
def _mat(u, name):
    return np.array(list(getattr(u, name)), dtype=np.float64).reshape(4, 4).T  # column-major -> math matrix


def test_camera_uniform_closed_form(oracle):
    cam_j = synth.orbit_cameras(8, 1200, 799, 1200.0, 1200.0)[3]
    cam = oracle.scene_camera_to_perspective(cam_j.position, cam_j.rotation, cam_j.fx, cam_j.fy, 1200, 799)
    cam.znear, cam.zfar = 0.5, 50.0
    u = oracle.camera_uniform(cam, 1200, 799)
    view, proj = _mat(u, "view"), _mat(u, "proj")
    pos = np.array(list(cam.position) + [1.0])
    assert np.allclose(view @ pos, [0, 0, 0, 1], atol=1e-5)                      # camera sits at the view origin
    c2w = np.array(cam_j.rotation, dtype=np.float64)                              # columns = camera axes
    assert np.allclose(view[:3, :3], c2w.T, atol=1e-5)                            # view rotation = world->camera
    assert np.allclose(_mat(u, "view_inv") @ view, np.eye(4), atol=1e-4)
    # the camera looks down +z at the origin: the target lands at the image centre with w = distance
    clip = proj @ view @ np.array([0, 0, 0, 1.0])
    assert abs(clip[0] / clip[3]) < 1e-5 and abs(clip[1] / clip[3]) < 1e-5
    assert np.isclose(clip[3], np.linalg.norm(cam_j.position), rtol=1e-5)
    # depth: z = zn -> 0, z = zf -> zf (clip space), and preprocess.wgsl:270-271 recovers znear / zfar
    for z, want in ((0.5, 0.0), (50.0, 50.0)):
        c = proj @ np.array([0, 0, z, 1.0])
        assert np.isclose(c[2], want, atol=1e-4)
    pm = np.array(list(u.proj), dtype=np.float32).reshape(4, 4)  # pm[c][r]
    assert np.isclose(-pm[3][2] / pm[2][2], 0.5, rtol=1e-5)
    assert np.isclose(-pm[3][2] / (pm[2][2] - 1.0), 50.0, rtol=1e-4)
    # y flip (camera.rs:107-112): a point BELOW the optical axis (camera y down, +y) has negative NDC y
    c = proj @ np.array([0.0, 1.0, 5.0, 1.0])
    assert c[1] / c[3] < 0
    # focal = viewport / (2 tan(fov/2)) gives back fx, fy
    assert np.isclose(u.focal[0], 1200.0, rtol=1e-5) and np.isclose(u.focal[1], 1200.0, rtol=1e-5)
    assert (u.viewport[0], u.viewport[1]) == (1200.0, 799.0)


def test_fit_near_far(oracle):
    cam = oracle.make_camera([0, 0, -10], [1, 0, 0, 0], 1.0, 1.0, 0.1, 100.0)
    bb = oracle.make_aabb([-1, -2, -3], [1, 2, 3])
    oracle.fit_near_far(cam, bb)
    r = np.linalg.norm([2, 4, 6]) / 2
    assert np.isclose(cam.zfar, 10 + r, rtol=1e-6)
    assert np.isclose(cam.znear, max(10 - r, (10 + r) / 1000), rtol=1e-6)
    cam2 = oracle.make_camera([0, 0, 0], [1, 0, 0, 0], 1.0, 1.0, 0.1, 100.0)  # inside the box: zfar/1000 floor
    oracle.fit_near_far(cam2, bb)
    assert np.isclose(cam2.znear, cam2.zfar / 1000.0, rtol=1e-6)


# ---- gaussian.wgsl:59-67 + blend state on hand-made splats ----------------------------------------------
def _splat(oracle, v1, v2, pos, rgba):
    h = [oracle.f32_to_f16(x) for x in (*v1, *v2, *pos, *rgba)]
    return np.array(h, dtype=np.uint16).view(np.uint8).reshape(1, 20)


def test_blend_single_isotropic_splat(oracle):
    W = H = 64
    sigma = 5.0  # px; eigenvalue lambda = sigma^2, v = sqrt(2 lambda) * e = sqrt(2) sigma
    s = np.sqrt(2.0) * sigma
    alpha = 0.8
    sp = _splat(oracle, (s / W, 0.0), (0.0, -s / H), (0.0, 0.0), (1.0, 0.5, 0.25, alpha))
    img = oracle.render(sp, None, W, H)
    dec = [oracle.f16_to_f32(oracle.f32_to_f16(x)) for x in (s / W, s / H, alpha, 0.5, 0.25)]
    sx, sy = dec[0] * W, dec[1] * H
    ys, xs = np.mgrid[0:H, 0:W]
    dx, dy = xs + 0.5 - W / 2, ys + 0.5 - H / 2
    a = (dx / sx) ** 2 + (dy / sy) ** 2  # = r^2 / (2 sigma^2)
    b = np.where(a <= 2 * 2.3539888583335364, np.minimum(0.99, np.exp(-a) * dec[2]), 0.0)
    assert np.allclose(img[..., 3], b, atol=2e-6)
    assert np.allclose(img[..., 0], b * 1.0, atol=2e-6)
    assert np.allclose(img[..., 1], b * dec[3], atol=2e-6)
    assert np.count_nonzero(img[..., 3]) == np.count_nonzero(a <= 2 * 2.3539888583335364)
    assert img[0, 0, 3] == 0.0  # exactly nothing outside the cut-off


def test_blend_alpha_clamp_and_order(oracle):
    W = H = 32
    s = 6.0
    near = _splat(oracle, (s / W, 0.0), (0.0, -s / H), (0.0, 0.0), (1.0, 0.0, 0.0, 1.0))   # red, opaque-ish
    far = _splat(oracle, (s / W, 0.0), (0.0, -s / H), (0.0, 0.0), (0.0, 0.0, 1.0, 1.0))    # blue
    both = np.concatenate([far, near])
    # draw order = sorted order: index 0 first (far), index 1 last (near)
    img = oracle.render(both, np.array([0, 1], dtype=np.uint32), W, H)
    c = img[H // 2, W // 2]
    bq = min(0.99, np.exp(-((0.5 / s) ** 2 * 2)) * 1.0)
    assert np.isclose(c[0], bq, atol=1e-5)                         # near colour at weight b
    assert np.isclose(c[2], bq * (1 - bq), atol=1e-5)              # far colour attenuated by (1 - b_near)
    assert np.isclose(c[3], 1 - (1 - bq) ** 2, atol=1e-5)
    assert bq == pytest.approx(0.99, abs=0.02)                     # clamp region at the centre
    # swapping the STORAGE order but keeping the draw order gives the same image
    img2 = oracle.render(np.concatenate([near, far]), np.array([1, 0], dtype=np.uint32), W, H)
    assert np.array_equal(img, img2)
    # background: dst starts as the clear colour and is attenuated by (1 - b) per splat
    img3 = oracle.render(both, np.array([0, 1], dtype=np.uint32), W, H, background=(0.2, 0.4, 0.6, 1.0))
    assert np.isclose(img3[0, 0, 1], 0.4) and np.isclose(img3[H // 2, W // 2, 1], 0.4 * (1 - bq) ** 2, atol=1e-5)


def test_target_modes_order(oracle):
    """f16 / unorm8 per-blend rounding stay close to the f32 target on a shallow stack."""
    rows = synth.scene_c1(n=2000, seed=5)
    g, sh = oracle.ply_rows_convert(rows, 3)
    bbox, center, _ = oracle.pointcloud_stats(g, 28, oracle.make_aabb([0, 0, 0], [0, 0, 0]))
    cj = synth.camera_c1(160, 120)
    cam = oracle.scene_camera_to_perspective(cj.position, cj.rotation, 160.0, 160.0, 160, 120)
    oracle.fit_near_far(cam, bbox)
    cu = oracle.camera_uniform(cam, 160, 120)
    rs = oracle.settings_uniform(bbox, center)
    splats, keys, _ = oracle.preprocess(g, sh, cu, rs)
    _, order = oracle.sort_pairs(keys, np.arange(len(keys), dtype=np.uint32))
    f32 = oracle.render(splats, order, 160, 120)
    f16 = oracle.render(splats, order, 160, 120, target_mode=1)
    u8 = oracle.render(splats, order, 160, 120, target_mode=2)
    assert f32[..., 3].max() > 0.5
    assert np.abs(f16 - f32).max() < 2e-2
    # a unorm8 target saturates at 1 (SH colours are not clamped above), so compare where nothing saturated
    ok = (f32.max(axis=-1) < 0.95)
    assert np.abs(u8 - f32)[ok].max() < 6e-2


def test_golden_sort_fixture(oracle):
    """tests/golden/sort_known_answer.json restates the reference's own vector (gpu_rs.rs:295-331)."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sort_known_answer.json")))
    n = g["n"]
    keys_in = np.arange(n - 1, -1, -1, dtype=np.float32).view(np.uint32)
    assert [int(x) for x in keys_in[:8]] == g["first8_in_bits"]
    k, _ = oracle.sort_pairs(keys_in, np.arange(n, dtype=np.uint32))
    assert [int(x) for x in k[:8]] == g["first8_out_bits"]
    assert np.array_equal(k.view(np.float32), np.arange(n, dtype=np.float32))


def test_snorm_times_127_is_a_clamp():
    """preprocess_compressed.wgsl:147-171: unpack4x8snorm(x) * 127 = max(x / 127, -1) * 127 in f32 returns every int8
    exactly (and -128 -> -127): the identity K1c relies on to drop the 48 divisions per Gaussian."""
    x = np.arange(-128, 128).astype(np.float32)
    y = (np.maximum(x / np.float32(127.0), np.float32(-1.0)) * np.float32(127.0)).astype(np.float32)
    assert np.array_equal(y, np.maximum(x, np.float32(-127.0)))


def test_sh_basis_is_the_real_spherical_harmonics(oracle):
    """Pins the oracle's SH evaluation (preprocess.wgsl:4-23 constants, 124-154 polynomials) against an independent
    source: scipy's spherical harmonics.  With a one-hot coefficient k = l^2 + l + m the colour minus 0.5 must be
    (up to the fixed sign convention of 3D Gaussian splatting) the real harmonic Y_lm of the view direction."""
    import scipy.special as sp

    def sph_harm(m, l, az, pol):  # complex Y_l^m; scipy renamed the function (and swapped the argument order) in 1.15
        if hasattr(sp, "sph_harm_y"):
            return sp.sph_harm_y(l, m, pol, az)
        return sp.sph_harm(m, l, az, pol)

    def real_sh(l, m, d):
        x, y, z = d
        az, pol = np.arctan2(y, x), np.arccos(z / np.linalg.norm(d))
        if m == 0:
            return float(sph_harm(0, l, az, pol).real)
        ylm = sph_harm(abs(m), l, az, pol)
        return float(np.sqrt(2.0) * (-1.0) ** m * (ylm.real if m > 0 else ylm.imag))

    amp = 0.25
    rng = np.random.default_rng(5)
    cam = oracle.make_camera([0.0, 0.0, -4.0], [1.0, 0.0, 0.0, 0.0], 0.9, 0.9, 0.1, 100.0, 1.0)
    cu = oracle.camera_uniform(cam, 400, 400)
    signs = {}
    for trial in range(6):
        pos = rng.uniform(-0.8, 0.8, size=3).astype(np.float32)
        d = pos.astype(np.float64) - np.array([0.0, 0.0, -4.0])
        d /= np.linalg.norm(d)
        n = 16
        g = np.zeros((n, 28), dtype=np.uint8)
        sh = np.zeros((n, 48), dtype=np.float16)
        for k in range(n):
            g[k, 0:12] = pos.view(np.uint8)
            g[k, 12:14] = np.array([1.0], dtype=np.float16).view(np.uint8)                 # opacity
            g[k, 16:28] = np.array([1e-4, 0, 0, 1e-4, 0, 1e-4], dtype=np.float16).view(np.uint8)  # small isotropic
            sh[k, 3 * k] = amp                                                              # red channel of coefficient k
        rs = oracle.settings_uniform(oracle.make_aabb([-1, -1, -1], [1, 1, 1]), [0.0, 0.0, 0.0], max_sh_deg=3)
        splats, keys, src = oracle.preprocess(g, sh.view(np.uint8).reshape(n, 96), cu, rs)
        assert len(keys) == n
        red = splats.view(np.float16).reshape(n, 10)[:, 6].astype(np.float64)[np.argsort(src)]
        for k in range(n):
            l = int(np.sqrt(k))
            m = k - l * l - l
            got = (red[k] - 0.5) / amp
            want = real_sh(l, m, d)
            assert abs(abs(got) - abs(want)) < 1.2e-2, (k, l, m, got, want)   # f16 colour: 1e-3 / amp
            if abs(want) > 0.1:
                signs.setdefault(k, set()).add(int(np.sign(got * want)))
    # the sign convention is a property of the basis, not of the direction; 3DGS carries the Condon-Shortley phase
    assert all(len(v) == 1 for v in signs.values()), signs
    assert signs[0] == {1} and signs[2] == {1} and signs[1] == {-1} and signs[3] == {-1}
