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


# ---- an independent float64 derivation of K1, as a second opinion on the C oracle -----------------------------------
def _k1_float64(gaussians, sh, cu, rs, sh_deg):
    """preprocess.wgsl:163-280 written from the mathematics (EWA projection of a 3D covariance, eigen-decomposition of
    the 2x2 screen covariance, real SH up to degree 3) in float64 numpy -- NOT from oracle/ws_oracle.c.
    Returns per Gaussian: visible mask, 10 floats of the Splat record (before f16 packing), the f32 depth key value."""
    g = np.ascontiguousarray(gaussians)
    n = g.shape[0]
    raw = g.view(np.uint8).reshape(n, 28)
    xyz = raw[:, 0:12].copy().view(np.float32).astype(np.float64)
    opacity = raw[:, 12:14].copy().view(np.float16).astype(np.float64)[:, 0]
    cov6 = raw[:, 16:28].copy().view(np.float16).astype(np.float64)            # xx xy xz yy yz zz
    coef = np.ascontiguousarray(sh).view(np.uint8).reshape(n, 96).copy().view(np.float16).astype(np.float64).reshape(n, 16, 3)
    return _k1_core_float64(xyz, opacity, cov6, coef, cu, rs, sh_deg, compressed=False)


def _k1_core_float64(xyz, opacity, cov6, coef, cu, rs, sh_deg, compressed):
    n = xyz.shape[0]
    V = np.array(cu.view[:], dtype=np.float64).reshape(4, 4).T                 # column-major storage -> matrix
    P = np.array(cu.proj[:], dtype=np.float64).reshape(4, 4).T
    Vinv = np.array(cu.view_inv[:], dtype=np.float64).reshape(4, 4).T
    fx, fy = float(cu.focal[0]), float(cu.focal[1])
    vw, vh = float(cu.viewport[0]), float(cu.viewport[1])
    cam = (V @ np.concatenate([xyz, np.ones((n, 1))], 1).T).T                  # camera space
    clip = (P @ cam.T).T
    w = clip[:, 3]
    z = clip[:, 2] / w
    bounds = 1.2 * w
    lo, hi = np.array(rs.clip_min[:3]), np.array(rs.clip_max[:3])
    vis = np.all(xyz >= lo, 1) & np.all(xyz <= hi, 1)
    zcull = ((z < 0) | (z > 1)) if compressed else ((z <= 0) | (z >= 1))   # preprocess_compressed.wgsl:229 vs preprocess.wgsl:192
    vis &= ~(zcull | (clip[:, 0] < -bounds) | (clip[:, 0] > bounds) | (clip[:, 1] < -bounds) | (clip[:, 1] > bounds))
    # fade-in
    dd = 5.0 * np.linalg.norm(np.array(rs.scene_center[:3]) - xyz, axis=1) / float(rs.scene_extend)
    t = np.clip(float(rs.walltime) - dd, 0.0, 1.0)
    scale_mod = np.where(float(rs.walltime) > dd, t * t * (3 - 2 * t), 0.0)
    s2 = (float(rs.gaussian_scaling) * scale_mod) ** 2
    S = np.empty((n, 3, 3))
    S[:, 0, 0], S[:, 0, 1], S[:, 0, 2] = cov6[:, 0], cov6[:, 1], cov6[:, 2]
    S[:, 1, 0], S[:, 1, 1], S[:, 1, 2] = cov6[:, 1], cov6[:, 3], cov6[:, 4]
    S[:, 2, 0], S[:, 2, 1], S[:, 2, 2] = cov6[:, 2], cov6[:, 4], cov6[:, 5]
    S *= s2[:, None, None]
    # screen-space Jacobian of (x, y) -> (fx x / z, -fy y / z) (the projection flips y) and the world->camera rotation
    x_, y_, z_ = cam[:, 0], cam[:, 1], cam[:, 2]
    Jm = np.zeros((n, 2, 3))
    Jm[:, 0, 0] = fx / z_
    Jm[:, 0, 2] = -fx * x_ / z_ ** 2
    Jm[:, 1, 1] = -fy / z_
    Jm[:, 1, 2] = fy * y_ / z_ ** 2
    R = V[:3, :3]
    A = Jm @ R                                                                   # d(screen) / d(world), n x 2 x 3
    cov2 = A @ S @ np.transpose(A, (0, 2, 1))
    ks = float(rs.kernel_size)
    a_, b_, c_ = cov2[:, 0, 0], cov2[:, 0, 1], cov2[:, 1, 1]
    op = opacity.copy()
    if int(rs.mip_splatting):
        det0 = np.maximum(1e-6, a_ * c_ - b_ * b_)
        det1 = np.maximum(1e-6, (a_ + ks) * (c_ + ks) - b_ * b_)
        k = np.sqrt(det0 / (det1 + 1e-6) + 1e-6)
        k = np.where((det0 <= 1e-6) | (det1 <= 1e-6), 0.0, k)
        op = op * k
    d1, d2 = a_ + ks, c_ + ks
    mid = 0.5 * (d1 + d2)
    rad = np.hypot((d1 - d2) / 2, b_)
    if compressed:   # preprocess_compressed.wgsl:296-297 clamps the radius, not the smaller eigenvalue
        l1, l2 = mid + np.maximum(rad, 0.1), mid - np.maximum(rad, 0.1)
    else:
        l1, l2 = mid + rad, np.maximum(mid - rad, 0.1)
    ex, ey = b_, l1 - d1                                                         # eigenvector of the larger eigenvalue
    nrm = np.hypot(ex, ey)
    ok = nrm > 0
    ex, ey = np.where(ok, ex / np.where(ok, nrm, 1), 1.0), np.where(ok, ey / np.where(ok, nrm, 1), 0.0)
    v1 = np.sqrt(2 * l1)[:, None] * np.stack([ex, ey], 1)
    with np.errstate(invalid="ignore"):
        v2 = np.sqrt(2 * l2)[:, None] * np.stack([ey, -ex], 1)
    centre = clip[:, :2] / w[:, None]
    campos = Vinv[:3, 3]
    d = xyz - campos
    d /= np.linalg.norm(d, axis=1)[:, None]
    x, y, zz_ = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    C0, C1 = 0.28209479177387814, 0.4886025119029199
    C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
    C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
          1.445305721320277, -0.5900435899266435]
    col = C0 * coef[:, 0]
    if sh_deg > 0:
        col = col - C1 * y * coef[:, 1] + C1 * zz_ * coef[:, 2] - C1 * x * coef[:, 3]
    if sh_deg > 1:
        xx, yy, z2, xy, yz, xz = x * x, y * y, zz_ * zz_, x * y, y * zz_, x * zz_
        col = col + C2[0] * xy * coef[:, 4] + C2[1] * yz * coef[:, 5] + C2[2] * (2 * z2 - xx - yy) * coef[:, 6] \
            + C2[3] * xz * coef[:, 7] + C2[4] * (xx - yy) * coef[:, 8]
    if sh_deg > 2:
        col = col + C3[0] * y * (3 * xx - yy) * coef[:, 9] + C3[1] * xy * zz_ * coef[:, 10] \
            + C3[2] * y * (4 * z2 - xx - yy) * coef[:, 11] + C3[3] * zz_ * (2 * z2 - 3 * xx - 3 * yy) * coef[:, 12] \
            + C3[4] * x * (4 * z2 - xx - yy) * coef[:, 13] + C3[5] * zz_ * (xx - yy) * coef[:, 14] \
            + C3[6] * x * (xx - 3 * yy) * coef[:, 15]
    col = np.maximum(col + 0.5, 0.0)
    rec = np.concatenate([v1 / [vw, vh], v2 / [vw, vh], centre, col, op[:, None]], 1)
    zfar = -P[2, 3] / (P[2, 2] - 1.0)
    if compressed:
        znear = -P[2, 3] / P[2, 2]
        return vis, rec, 16777215.0 - (clip[:, 2] - znear) / (zfar - znear) * 16777215.0
    return vis, rec, zfar - clip[:, 2]


@pytest.mark.parametrize("mip,sh_deg", [(False, 3), (True, 3), (False, 1)])
def test_k1_oracle_agrees_with_float64_derivation(oracle, mip, sh_deg):
    """The C oracle (f32, WGSL operation order) against an independent float64 derivation of the same mathematics:
    same visible set (up to Gaussians within 1e-5 of a cull plane), every f16 field of the Splat record within
    2 f16 ulp + the f32-vs-f64 slack of the eigen-decomposition, depth keys within 1e-5 relative."""
    from websplat import synth
    rows = synth.scene_c1(n=4000, seed=31, sh_deg=3)
    g, sh = oracle.ply_rows_convert(rows, 3)
    cj = synth.camera_c1(800, 600)
    cam = oracle.scene_camera_to_perspective(cj.position, cj.rotation, cj.fx, cj.fy, 800, 600)
    pos = g.view(np.uint8).reshape(-1, 28)[:, :12].copy().view(np.float32)
    aabb = oracle.make_aabb(pos.min(0), pos.max(0))
    oracle.fit_near_far(cam, aabb)
    cu = oracle.camera_uniform(cam, 800, 600)
    rs = oracle.settings_uniform(aabb, pos.mean(0), max_sh_deg=sh_deg, mip_splatting=mip, kernel_size=0.3)
    splats, keys, src = oracle.preprocess(g, sh, cu, rs)
    vis, rec, key64 = _k1_float64(g, sh, cu, rs, sh_deg)
    mine = set(np.nonzero(vis)[0].tolist())
    theirs = set(src.tolist())
    assert len(mine ^ theirs) <= 2, sorted(mine ^ theirs)[:10]
    both = np.array(sorted(mine & theirs))
    assert len(both) > 1000
    row_of = {int(s): i for i, s in enumerate(src.tolist())}
    got = splats[[row_of[int(i)] for i in both]].view(np.float16).astype(np.float64).reshape(-1, 10)
    want = rec[both]
    # an eigenvector is defined up to sign: the reference's choice (offDiagonal, lambda1 - d1) is reproduced above,
    # so no sign fix-up is needed; near-isotropic splats (radius ~ 0) have an ill-conditioned direction -> compare the
    # covariance v1 v1^T + v2 v2^T instead of the vectors for those
    sig_got = np.einsum("ni,nj->nij", got[:, 0:2], got[:, 0:2]) + np.einsum("ni,nj->nij", got[:, 2:4], got[:, 2:4])
    sig_want = np.einsum("ni,nj->nij", want[:, 0:2], want[:, 0:2]) + np.einsum("ni,nj->nij", want[:, 2:4], want[:, 2:4])
    scale = np.abs(sig_want).max(axis=(1, 2))[:, None, None]
    assert np.max(np.abs(sig_got - sig_want) / scale) < 6e-3                       # two f16 roundings of each factor
    aniso = np.abs(np.linalg.norm(want[:, 0:2], axis=1) / np.linalg.norm(want[:, 2:4], axis=1) - 1) > 0.05
    v_err = np.abs(got[aniso, 0:4] - want[aniso, 0:4]) / np.abs(want[aniso, 0:4]).max(axis=1)[:, None]
    assert np.percentile(v_err, 99) < 4e-3 and v_err.max() < 5e-2
    assert np.max(np.abs(got[:, 4:6] - want[:, 4:6])) < 1.5e-3                     # NDC centre: f16 at |x| <= 1.2
    assert np.max(np.abs(got[:, 6:10] - want[:, 6:10]) / np.maximum(np.abs(want[:, 6:10]), 0.05)) < 3e-3
    k_got = keys[[row_of[int(i)] for i in both]].view(np.float32).astype(np.float64)
    assert np.max(np.abs(k_got - key64[both]) / np.abs(key64[both])) < 1e-5


def test_render_oracle_agrees_with_float64_derivation(oracle):
    """gaussian.wgsl:29-66 + PREMULTIPLIED_ALPHA_BLENDING (renderer.rs:63-67) derived independently in float64:
    the quad vertex stage makes screen_pos the linear map (2 [v1 v2])^-1 (p_ndc - centre) of a pixel centre, the
    fragment stage keeps a = |screen_pos|^2 <= 2*CUTOFF with weight min(0.99, exp(-a) alpha), drawn far -> near with
    dst = src + dst (1 - src.a) over the cleared target."""
    from websplat import synth
    W, H = 96, 64
    rows = synth.scene_c1(n=400, seed=41, sh_deg=3)
    rows[:, -7:-4] += 1.2                                        # larger splats: plenty of overlap per pixel
    g, sh = oracle.ply_rows_convert(rows, 3)
    cj = synth.camera_c1(W, H)
    cam = oracle.scene_camera_to_perspective(cj.position, cj.rotation, 90.0, 90.0, W, H)
    pos = g.view(np.uint8).reshape(-1, 28)[:, :12].copy().view(np.float32)
    aabb = oracle.make_aabb(pos.min(0), pos.max(0))
    oracle.fit_near_far(cam, aabb)
    cu = oracle.camera_uniform(cam, W, H)
    rs = oracle.settings_uniform(aabb, pos.mean(0))
    splats, keys, _ = oracle.preprocess(g, sh, cu, rs)
    _, order = oracle.sort_pairs(keys, np.arange(len(keys), dtype=np.uint32))
    bg = (0.1, 0.2, 0.3, 1.0)
    img = oracle.render(splats, order, W, H, bg, 0).astype(np.float64)

    s = splats.view(np.float16).astype(np.float64).reshape(-1, 10)
    py, px = np.mgrid[0:H, 0:W]
    ndc = np.stack([(px + 0.5) / W * 2 - 1, 1 - (py + 0.5) / H * 2], -1)   # framebuffer y points down
    out = np.broadcast_to(np.array(bg, dtype=np.float64), (H, W, 4)).copy()
    cut = 2 * 2.3539888583335364
    stable = np.argsort(keys, kind="stable")                                 # far -> near, ties in store order
    assert np.array_equal(stable.astype(np.uint32), order)
    for i in stable:
        M = 2.0 * np.array([[s[i, 0], s[i, 2]], [s[i, 1], s[i, 3]]])         # columns v1, v2
        det = np.linalg.det(M)
        if det == 0 or not np.isfinite(det):
            continue
        sp = (ndc - s[i, 4:6]) @ np.linalg.inv(M).T
        a = (sp ** 2).sum(-1)
        b = np.where(a <= cut, np.minimum(0.99, np.exp(-a) * s[i, 9]), 0.0)[..., None]
        src = np.concatenate([s[i, 6:9] * b, b], -1)
        out = src + out * (1 - b)
    # a fragment whose `a` is within rounding of the cut-off may be kept by one evaluation and discarded by the
    # other (weight step <= 0.009): allow a handful of such pixels, everything else agrees to f32 accumulation error
    diff = np.abs(img - out).max(-1)
    assert np.mean(diff) < 2e-6
    assert (diff > 5e-5).sum() <= 4 and diff.max() < 0.0135


@pytest.mark.parametrize("sh_deg", [3, 1])
def test_k1c_oracle_agrees_with_float64_derivation(oracle, sh_deg):
    """preprocess_compressed.wgsl:137-332 (int8 de-quantisation, codebook covariance x exp(scale)^2, packed SH records,
    24-bit depth key, the radius-clamped eigenvalues) derived independently in float64 against the C oracle."""
    from websplat import synth
    blobs = synth.compressed_blobs(n=6000, n_geometry=256, n_sh=199, seed=51, sh_deg=sh_deg)
    gdt = np.dtype([("xyz", "<f4", 3), ("opacity", "i1"), ("scale_factor", "i1"), ("pad", "u1", 2),
                    ("geometry_idx", "<u4"), ("sh_idx", "<u4")])             # GaussianCompressed, pointcloud.rs:14-22
    g = np.ascontiguousarray(blobs["gaussians"]).view(np.uint8).reshape(-1, 24).copy().view(gdt).reshape(-1)
    n = len(g)
    q = blobs["quant"]
    xyz = g["xyz"].astype(np.float64)
    deq = lambda v, name: (v.astype(np.float64) - q[name][0]) * np.float64(np.float32(q[name][1]))
    opacity = deq(g["opacity"], "opacity")
    s2 = np.exp(deq(g["scale_factor"], "scaling_factor")) ** 2
    cov6 = blobs["covars"].view(np.float16).reshape(-1, 6).astype(np.float64)[g["geometry_idx"]] * s2[:, None]
    ncoef = (sh_deg + 1) ** 2
    sh8 = np.ascontiguousarray(blobs["sh"]).view(np.int8).reshape(-1, 3 * ncoef)[g["sh_idx"]].astype(np.float64)
    sh8 = np.maximum(sh8, -127.0)                                   # unpack4x8snorm clamps -128 to -1.0
    coef = np.zeros((n, 16, 3))
    coef[:, 0] = (sh8[:, 0:3] - q["color_dc"][0]) * np.float64(np.float32(q["color_dc"][1]))
    coef[:, 1:ncoef] = ((sh8[:, 3:] - q["color_rest"][0]) * np.float64(np.float32(q["color_rest"][1]))).reshape(n, ncoef - 1, 3)
    cj = synth.look_at_camera(0, [0.0, 0.0, -3.0], [0, 0, 0], 800, 600, 800.0, 800.0)
    cam = oracle.scene_camera_to_perspective(cj.position, cj.rotation, cj.fx, cj.fy, 800, 600)
    aabb = oracle.make_aabb(xyz.min(0), xyz.max(0))
    oracle.fit_near_far(cam, aabb)
    cu = oracle.camera_uniform(cam, 800, 600)
    rs = oracle.settings_uniform(aabb, xyz.mean(0), max_sh_deg=sh_deg)
    oq = oracle.make_quantization(q)
    splats, keys, src = oracle.preprocess_compressed(blobs["gaussians"], blobs["sh"], blobs["covars"], oq, sh_deg, cu, rs)
    vis, rec, key64 = _k1_core_float64(xyz, opacity, cov6, coef, cu, rs, sh_deg, compressed=True)
    mine, theirs = set(np.nonzero(vis)[0].tolist()), set(src.tolist())
    assert len(mine ^ theirs) <= 2, sorted(mine ^ theirs)[:10]
    both = np.array(sorted(mine & theirs))
    assert len(both) > 1000
    row_of = {int(s_): i for i, s_ in enumerate(src.tolist())}
    rows = [row_of[int(i)] for i in both]
    got = splats[rows].view(np.float16).astype(np.float64).reshape(-1, 10)
    want = rec[both]
    fin = np.all(np.isfinite(want), 1) & np.all(np.isfinite(got), 1)      # lambda2 < 0 gives NaN axes on both sides
    assert np.array_equal(np.all(np.isfinite(want), 1), np.all(np.isfinite(got), 1))
    assert fin.sum() > 1000
    got, want = got[fin], want[fin]
    sig = lambda v: np.einsum("ni,nj->nij", v[:, 0:2], v[:, 0:2]) + np.einsum("ni,nj->nij", v[:, 2:4], v[:, 2:4])
    scale = np.abs(sig(want)).max(axis=(1, 2))[:, None, None]
    assert np.max(np.abs(sig(got) - sig(want)) / scale) < 6e-3
    assert np.max(np.abs(got[:, 4:6] - want[:, 4:6])) < 1.5e-3
    assert np.max(np.abs(got[:, 6:10] - want[:, 6:10]) / np.maximum(np.abs(want[:, 6:10]), 0.05)) < 3e-3
    k_got = keys[rows].astype(np.float64)[fin]
    # 24-bit integer keys: the f32 expression carries a few ulp (1-2 units at 1.6e7 each) into the truncation
    assert np.max(np.abs(k_got - np.floor(key64[both][fin]))) <= 4
