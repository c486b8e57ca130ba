#!/usr/bin/env python3
"""CPU study (no GPU): how many (tile, splat) entries does binning by the kept ELLIPSE produce, against binning by
its bounding box?  Runs the oracle's K1 on a bench workload's view 0 and counts, per visible splat, the binning tiles
(a) of the bounding rectangle K1 derives today and (b) that contain at least one point of the kept ellipse
a <= 2*CUTOFF restricted to the tile's pixel-centre box (the band-exact span of blend_stage.h, in float64).

  python scripts/footprint_study.py [hd1m|c2|c3|c4] [tile_px]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "web-splat_amd"), os.path.join(ROOT, "tests"), ROOT]

CUT_A = 2.0 * 2.3539888583335364


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "hd1m"
    tile = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    import oracle_lib as oracle
    from websplat import synth
    # host-side scene construction without the HIP library: the oracle's own converters
    if name == "c3":
        rows, (w, h) = synth.scene_c3(n=5_000_000, seed=2), (1920, 1080)
        cj = synth.camera_c3(w, h)
    else:
        n = 1_000_000 if name == "hd1m" else 1_200_000
        (w, h), f = ((1200, 799), 1200.0) if name == "c2" else ((1920, 1080), 1920.0)
        rows = synth.scene_c2(n=n, seed=1)
        cj = synth.orbit_cameras(64, w, h, f, f)[0]
    g, sh = oracle.ply_rows_convert(rows, 3)
    start = oracle.Aabb()
    start.min[:] = [np.inf] * 3
    start.max[:] = [-np.inf] * 3
    bbox, center, _ = oracle.pointcloud_stats(g, 28, start)
    cam = oracle.scene_camera_to_perspective(cj.position, cj.rotation, cj.fx, cj.fy, cj.width, cj.height)
    oracle.fit_near_far(cam, bbox)
    cu = oracle.camera_uniform(cam, w, h)
    rs = oracle.settings_uniform(bbox, center)
    splats, keys, _ = oracle.preprocess(g, sh, cu, rs)
    hv = np.ascontiguousarray(splats).view(np.float16).reshape(-1, 10).astype(np.float64)
    W, H = float(w), float(h)
    m00, m01 = hv[:, 0] * W, hv[:, 2] * W
    m10, m11 = -hv[:, 1] * H, -hv[:, 3] * H
    det = m00 * m11 - m01 * m10
    ok = np.isfinite(det) & (np.abs(det) > 0)
    cx = (hv[:, 4] * 0.5 + 0.5) * W
    cy = (0.5 - hv[:, 5] * 0.5) * H
    rad = np.sqrt(CUT_A)
    exx = rad * np.sqrt(m00 ** 2 + m01 ** 2)
    eyy = rad * np.sqrt(m10 ** 2 + m11 ** 2)
    x_lo = np.maximum(np.ceil(cx - exx - 0.5), 0)
    x_hi = np.minimum(np.floor(cx + exx - 0.5), W - 1)
    y_lo = np.maximum(np.ceil(cy - eyy - 0.5), 0)
    y_hi = np.minimum(np.floor(cy + eyy - 0.5), H - 1)
    vis = ok & (x_lo <= x_hi) & (y_lo <= y_hi)
    tx0, tx1 = (x_lo // tile).astype(np.int64), (x_hi // tile).astype(np.int64)
    ty0, ty1 = (y_lo // tile).astype(np.int64), (y_hi // tile).astype(np.int64)
    bw, bh = np.where(vis, tx1 - tx0 + 1, 0), np.where(vis, ty1 - ty0 + 1, 0)
    bbox_cnt = bw * bh
    # exact: inverse map I = M^-1; a(d) = A dx^2 + B2 dx dy + C dy^2
    inv = 1.0 / np.where(ok, det, 1.0)
    i00, i01, i10, i11 = m11 * inv, -m01 * inv, -m10 * inv, m00 * inv
    A = i00 ** 2 + i10 ** 2
    C = i01 ** 2 + i11 ** 2
    B2 = 2 * (i00 * i01 + i10 * i11)
    D = A * C - B2 ** 2 / 4
    ymax = np.sqrt(CUT_A * A / D)
    xmax = np.sqrt(CUT_A * C / D)
    k = -0.5 * B2 / A
    ys = -0.5 * B2 / C * xmax
    exact = np.zeros(len(hv), dtype=np.int64)
    maxh = int(bh.max())
    idx = np.nonzero(vis)[0]
    for r in range(maxh):
        sel = idx[bh[idx] > r]
        if len(sel) == 0:
            break
        ty = ty0[sel] + r
        # pixel centres of the tile row, clipped to the image
        y0 = ty * tile + 0.5 - cy[sel]
        y1 = np.minimum(ty * tile + tile - 1, H - 1) + 0.5 - cy[sel]
        lo, hi = np.maximum(y0, -ymax[sel]), np.minimum(y1, ymax[sel])
        has = lo <= hi
        yr, yl = np.clip(ys[sel], lo, hi), np.clip(-ys[sel], lo, hi)
        cA, dA2 = CUT_A / A[sel], D[sel] / A[sel] ** 2
        sr = np.sqrt(np.maximum(cA - dA2 * yr ** 2, 0))
        sl = np.sqrt(np.maximum(cA - dA2 * yl ** 2, 0))
        x1 = cx[sel] + k[sel] * yr + sr
        x0 = cx[sel] + k[sel] * yl - sl
        # pixel columns whose centre lies in [x0, x1], clipped to the image, then tiles
        px0 = np.maximum(np.ceil(x0 - 0.5), 0)
        px1 = np.minimum(np.floor(x1 - 0.5), W - 1)
        has &= px0 <= px1
        c = np.where(has, px1 // tile - px0 // tile + 1, 0).astype(np.int64)
        exact[sel] += c
    # (c) the reference's ORIENTED QUAD (gaussian.wgsl:40-53: centre +- cut * v1 +- cut * v2, cut = sqrt(2 CUTOFF)), exact
    #     polygon-vs-band x-range: no square roots per row, but looser than the ellipse (area 4 / pi)
    r = np.sqrt(CUT_A)
    a1x, a1y, a2x, a2y = m00 * r, m10 * r, m01 * r, m11 * r
    vx = np.stack([cx + a1x + a2x, cx + a1x - a2x, cx - a1x - a2x, cx - a1x + a2x], 1)
    vy = np.stack([cy + a1y + a2y, cy + a1y - a2y, cy - a1y - a2y, cy - a1y + a2y], 1)
    quad = np.zeros(len(hv), dtype=np.int64)
    for rr in range(maxh):
        sel = idx[bh[idx] > rr]
        if len(sel) == 0:
            break
        ty = ty0[sel] + rr
        y0 = (ty * tile + 0.5)[:, None]
        y1 = (np.minimum(ty * tile + tile - 1, H - 1) + 0.5)[:, None]
        X, Y = vx[sel], vy[sel]
        lo = np.full(len(sel), np.inf)
        hi = np.full(len(sel), -np.inf)
        inside = (Y >= y0) & (Y <= y1)
        lo = np.minimum(lo, np.where(inside, X, np.inf).min(1))
        hi = np.maximum(hi, np.where(inside, X, -np.inf).max(1))
        for e in range(4):
            xa, ya, xb, yb = X[:, e], Y[:, e], X[:, (e + 1) % 4], Y[:, (e + 1) % 4]
            for yl in (y0[:, 0], y1[:, 0]):
                with np.errstate(all="ignore"):
                    t = (yl - ya) / (yb - ya)
                ok_ = np.isfinite(t) & (t >= 0) & (t <= 1)
                xi = xa + t * (xb - xa)
                lo = np.minimum(lo, np.where(ok_, xi, np.inf))
                hi = np.maximum(hi, np.where(ok_, xi, -np.inf))
        px0 = np.maximum(np.ceil(lo - 0.5), 0)
        px1 = np.minimum(np.floor(hi - 0.5), W - 1)
        has = np.isfinite(lo) & (px0 <= px1)
        c = np.where(has, np.minimum(px1 // tile, tx1[sel]) - np.maximum(px0 // tile, tx0[sel]) + 1, 0).astype(np.int64)
        quad[sel] += np.maximum(c, 0)
    print(f"D oriented quad {int(quad.sum())} ({quad.sum() / max(bbox_cnt.sum(), 1):.3f} of bbox)")
    V = int(vis.sum())
    Db, De = int(bbox_cnt.sum()), int(exact.sum())
    print(f"{name} {w}x{h} tile {tile}: visible {len(hv)}, with tiles {V}; D bbox {Db} ({Db / max(len(hv), 1):.2f}/splat), "
          f"D ellipse {De} ({De / Db:.3f} of bbox)")
    # where do the entries come from?  by bounding-rectangle size
    print("rect size class: splats, bbox entries, ellipse entries")
    for name_, m in (("1x1", (bw == 1) & (bh == 1)), ("<=2x2", (bw <= 2) & (bh <= 2) & ~((bw == 1) & (bh == 1))),
                     ("<=4x4", (bw <= 4) & (bh <= 4) & ~((bw <= 2) & (bh <= 2))),
                     ("<=8x8", (bw <= 8) & (bh <= 8) & ~((bw <= 4) & (bh <= 4))),
                     (">8x8", (bw > 8) | (bh > 8))):
        m = m & vis
        print(f"  {name_:6s} {int(m.sum()):9d} {int(bbox_cnt[m].sum()):10d} {int(exact[m].sum()):10d}")
    zero = vis & (exact == 0)
    print(f"splats whose ellipse reaches no pixel centre at all: {int(zero.sum())}")


if __name__ == "__main__":
    main()
