"""CPU tests of the parity tooling itself (tests/scenes.py): the cut-off boundary allowance has to PROVE that a
forgiven pixel is a boundary pixel (BoundaryProof), and must reject a pixel that is simply wrong."""
import numpy as np

import scenes


def _splat_record(v1, v2, centre_ndc, rgba):
    return np.array([*v1, *v2, *centre_ndc, *rgba], dtype=np.float16).view(np.uint8)


def _find_boundary_configuration(W, H):
    """An isotropic splat (v1 = (s/W, 0), v2 = (0, s/H) with s/W an f16 value) and an integer pixel offset (dx, dy)
    whose a = (dx^2 + dy^2) / s^2 is within 1e-7 of the cut-off."""
    best = None
    for dx in range(1, 110):
        for dy in range(0, dx + 1):
            n = dx * dx + dy * dy
            s_star = np.sqrt(n / scenes.CUT_A)
            if not (3.0 < s_star < 50.0):
                continue
            q0 = np.float16(s_star / W)
            for q in (np.nextafter(q0, np.float16(0)), q0, np.nextafter(q0, np.float16(1))):
                err = abs(n / (float(q) * W) ** 2 - scenes.CUT_A)
                if best is None or err < best[0]:
                    best = (err, np.float16(q), dx, dy)
    return best


def test_boundary_proof_accepts_a_true_boundary_pixel_and_rejects_a_wrong_one(oracle):
    W, H = 256, 256
    err, q, ddx, ddy = _find_boundary_configuration(W, H)
    assert err < 5e-6, err
    # splat A: the boundary splat, centred on the centre of pixel (100, 120); splat B: a plain one covering everything nearby
    cx, cy = 100.5, 120.5
    ndc = (cx / W * 2 - 1, 1 - cy / H * 2)
    qh = np.float16(float(q) * W / H)
    rec_a = _splat_record((q, 0), (0, qh), ndc, (0.9, 0.5, 0.2, 0.8))
    rec_b = _splat_record((np.float16(0.2), 0), (0, np.float16(0.2)), ndc, (0.1, 0.3, 0.7, 0.5))
    splats = np.stack([rec_b, rec_a])
    order = np.array([0, 1], dtype=np.uint32)          # B far, A near
    ref = oracle.render(splats, order, W, H, (0, 0, 0, 0), 0)
    proof = scenes.BoundaryProof(splats, order, W, H)
    x, y = 100 + ddx, 120 + ddy                        # a = |pixel - centre|^2 / s^2 ~ cut-off
    a_vals = ((x + 0.5 - proof.cx) * proof.i00) ** 2 + ((y + 0.5 - proof.cy) * proof.i11) ** 2
    assert abs(a_vals[1] - scenes.CUT_A) < 1e-5
    # the two legitimate values of that pixel: fragment A kept / discarded
    b_b = min(0.99, np.exp(-a_vals[0]) * 0.5)
    col_b = np.array([0.1, 0.3, 0.7, 1.0], dtype=np.float16).astype(np.float64)
    col_a = np.array([0.9, 0.5, 0.2, 1.0], dtype=np.float16).astype(np.float64)
    b_a = min(0.99, np.exp(-scenes.CUT_A) * float(np.float16(0.8)))
    only_b = col_b * b_b
    with_a = col_a * b_a + only_b * (1 - b_a)
    assert np.abs(with_a - only_b).max() > scenes.MAX_ABS          # the flip is visible above the plain tolerance
    for value in (only_b, with_a):
        ok, msg = proof.explain(x, y, value)
        assert ok, msg
    # the oracle itself took one of the two
    assert min(np.abs(ref[y, x] - only_b).max(), np.abs(ref[y, x] - with_a).max()) < 1e-5
    # a wrong value at the boundary pixel, and a dropped fragment at a NON-boundary pixel, are rejected
    ok, msg = proof.explain(x, y, with_a + 0.01)
    assert not ok and "no keep/discard assignment" in msg
    ok, msg = proof.explain(x - 3, y, ref[y, x - 3] * 0.5)
    assert not ok and "not a boundary pixel" in msg
    # image_close with the proof: a flipped boundary pixel passes, the same error elsewhere fails
    img = ref.copy()
    img[y, x] = with_a if np.abs(ref[y, x] - only_b).max() < 1e-5 else only_b
    assert scenes.image_close(img, ref, proof=proof)[0]
    img2 = ref.copy()
    img2[y, x - 3] += 0.008
    ok, msg, *_ = scenes.image_close(img2, ref, proof=proof)
    assert not ok and "not a boundary pixel" in msg
    assert scenes.image_close(img2, ref)[0]            # ... which the unproven allowance would have forgiven
