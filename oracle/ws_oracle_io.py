"""Oracle (test infrastructure only, like ws_oracle.c): numpy restatement of the reference's host-side data
preparation around the hot path.  Imported by tests/ only; the product path never touches it.

  npz_decode            src/io/npz.rs:59-225  (NpzReader::read) with numpy's own .npz reader as the container parser
  scene_from_json       src/scene.rs:113-194  (Scene::from_json / cameras / extend / nearest_camera)
  download_texture_u8   src/bin/render.rs:222-236 (f16 -> clamp -> *255 -> `as u8`)
  display_composite     src/renderer.rs:548-582 + src/shaders/display.wgsl:37-55 + PREMULTIPLIED_ALPHA_BLENDING
  png_read_rgba8        reads back an RGBA8 PNG (ISO/IEC 15948; all five filter types) to check the writer

parity unpinned: the reference ships no fixtures for these paths; the restatement follows the cited
lines and is pinned only against closed forms in tests/test_io.py.
"""
import json
import struct
import zlib

import numpy as np

F = np.float32


def _scalar(z, name, default, kind):
    if name not in z.files:
        return default
    v = np.asarray(z[name]).reshape(-1)
    if v.size == 0:
        raise ValueError("array empty")
    return kind(v[0])


def _quat_to_mat3(q):
    """cgmath `impl From<Quaternion<S>> for Matrix3<S>`, q = (s, x, y, z), f32 in source order; columns c0, c1, c2."""
    s, x, y, z = (q[:, i] for i in range(4))
    x2, y2, z2 = x + x, y + y, z + z
    xx2, xy2, xz2 = x2 * x, x2 * y, x2 * z
    yy2, yz2, zz2 = y2 * y, y2 * z, z2 * z
    sy2, sz2, sx2 = y2 * s, z2 * s, x2 * s
    one = F(1.0)
    c0 = np.stack([one - yy2 - zz2, xy2 + sz2, xz2 - sy2], 1)
    c1 = np.stack([xy2 - sz2, one - xx2 - zz2, yz2 + sx2], 1)
    c2 = np.stack([xz2 + sy2, yz2 - sx2, one - xx2 - yy2], 1)
    return [c0, c1, c2]


def build_cov(q, scale):
    """utils.rs:194-203: l = R * diag(s); m = l * l^T; returns [m00, m01, m02, m11, m12, m22] (f32)."""
    cols = _quat_to_mat3(q.astype(F))
    sc = scale.astype(F)
    zero = np.zeros_like(sc[:, 0:1])
    l = []
    for c in range(3):  # column c of R * from_diagonal(s): row.dot(column) = x*x' + y*y' + z*z', zero terms included
        d = [sc[:, k:k + 1] if k == c else zero for k in range(3)]
        l.append((cols[0] * d[0] + cols[1] * d[1]) + cols[2] * d[2])

    def m(c, r):  # element (column c, row r) of l * l^T = sum_k l[k][r] * l[k][c]
        acc = l[0][:, r] * l[0][:, c]
        acc = acc + l[1][:, r] * l[1][:, c]
        acc = acc + l[2][:, r] * l[2][:, c]
        return acc
    return np.stack([m(0, 0), m(0, 1), m(0, 2), m(1, 1), m(1, 2), m(2, 2)], 1).astype(F)


def npz_decode(path):
    z = np.load(path)
    sh_deg = 0
    if "features_rest" in z.files:
        ncoef = z["features_rest"].shape[1] + 1
        root = int(round(np.sqrt(ncoef)))
        if root * root != ncoef:
            raise ValueError("num sh coefs not valid")
        sh_deg = root - 1
    opacity_scale = _scalar(z, "opacity_scale", F(1.0), F)
    opacity_zp = _scalar(z, "opacity_zero_point", 0, int)
    scaling_scale = _scalar(z, "scaling_scale", F(1.0), F)
    scaling_zp = F(_scalar(z, "scaling_zero_point", 0, int))
    rotation_scale = _scalar(z, "rotation_scale", F(1.0), F)
    rotation_zp = F(_scalar(z, "rotation_zero_point", 0, int))
    dc_scale = _scalar(z, "features_dc_scale", F(1.0), F)
    dc_zp = _scalar(z, "features_dc_zero_point", 0, int)
    rest_scale = _scalar(z, "features_rest_scale", F(1.0), F)
    rest_zp = _scalar(z, "features_rest_zero_point", 0, int)
    has_sf = "scaling_factor_scale" in z.files
    sf_scale, sf_zp, sf = F(1.0), 0, None
    if has_sf:
        sf_scale = _scalar(z, "scaling_factor_scale", F(1.0), F)
        sf_zp = _scalar(z, "scaling_factor_zero_point", 0, int)
        sf = z["scaling_factor"].reshape(-1).astype(np.int8)
    xyz = z["xyz"].reshape(-1, 3).astype(np.float16).astype(F)
    n = xyz.shape[0]
    sc = (z["scaling"].reshape(-1).astype(F) - scaling_zp) * scaling_scale
    if has_sf:
        sc = np.maximum(sc, F(0.0)).reshape(-1, 3)
        mag = np.sqrt(sc[:, 0] * sc[:, 0] + sc[:, 1] * sc[:, 1] + sc[:, 2] * sc[:, 2])
        sc = sc * (F(1.0) / mag)[:, None]
    else:
        sc = np.exp(sc).astype(F).reshape(-1, 3)
    q = ((z["rotation"].reshape(-1).astype(F) - rotation_zp) * rotation_scale).reshape(-1, 4)
    mag = np.sqrt(q[:, 0] * q[:, 0] + (q[:, 1] * q[:, 1] + q[:, 2] * q[:, 2] + q[:, 3] * q[:, 3]))
    q = q * (F(1.0) / mag)[:, None]
    covars = build_cov(q, sc).astype(np.float16)
    g = np.zeros(n, dtype=np.dtype([("xyz", "<f4", 3), ("opacity", "i1"), ("scale_factor", "i1"), ("pad", "u1", 2),
                                    ("geometry_idx", "<u4"), ("sh_idx", "<u4")]))
    g["xyz"] = xyz
    g["opacity"] = z["opacity"].reshape(-1)[:n]
    g["scale_factor"] = sf[:n] if has_sf else 0
    g["geometry_idx"] = z["gaussian_indices"].reshape(-1).astype(np.uint32) if "gaussian_indices" in z.files else np.arange(n, dtype=np.uint32)
    g["sh_idx"] = z["feature_indices"].reshape(-1).astype(np.uint32) if "feature_indices" in z.files else np.arange(n, dtype=np.uint32)
    dc = z["features_dc"].reshape(-1, 3).astype(np.int8)
    ncoef = (sh_deg + 1) ** 2
    rest = z["features_rest"].reshape(dc.shape[0], ncoef * 3 - 3).astype(np.int8)
    sh = np.concatenate([dc, rest], axis=1)
    out = {
        "gaussians": g.view(np.uint8).reshape(n, 24), "sh": sh.view(np.uint8).reshape(-1),
        "covars": covars.view(np.uint8).reshape(-1, 12), "sh_deg": sh_deg, "num_points": n,
        "quant": {"color_dc": (dc_zp, dc_scale), "color_rest": (rest_zp, rest_scale),
                  "opacity": (opacity_zp, opacity_scale), "scaling_factor": (sf_zp, sf_scale)},
        "kernel_size": _scalar(z, "kernel_size", None, F), "mip_splatting": _scalar(z, "mip_splatting", None, bool),
        "background_color": [F(v) for v in z["background_color"].reshape(-1)] if "background_color" in z.files else None,
    }
    return out


def scene_from_json(text):
    cams = json.loads(text)
    for i, c in enumerate(cams):
        c["split"] = "test" if i % 8 == 0 else "train"
    pos = np.array([c["position"] for c in cams], dtype=F).reshape(-1, 3)
    max_d2 = F(0.0)
    for i in range(len(cams)):
        for j in range(i + 1, len(cams)):
            d = pos[i] - pos[j]
            max_d2 = max(max_d2, F(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]))
    by_id = {}
    for c in cams:
        by_id[c["id"]] = c
    return {"cameras": by_id, "extend": F(np.sqrt(max_d2))}


def scene_cameras(scene, split=None):
    return [c for _, c in sorted(scene["cameras"].items()) if split is None or c["split"] == split]


def scene_nearest(scene, p, split=None):
    best = None
    for cid, c in sorted(scene["cameras"].items()):
        if split is not None and c["split"] != split:
            continue
        d = np.array(c["position"], dtype=F) - np.array(p, dtype=F)
        key = min(int(F(F(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) * F(1e6))), 0xFFFFFFFF)
        if best is None or key < best[0]:
            best = (key, cid)
    return None if best is None else best[1]


def download_texture_u8(img):
    v = np.clip(img.astype(F), F(0.0), F(1.0)) * F(255.0)
    return v.astype(np.uint8)  # truncation, `as u8`


def display_composite(img, background, bgra=False):
    src = img.astype(F)
    k = F(1.0) - src[..., 3:4]
    out = src + np.array(background, dtype=F) * k
    q = np.rint(np.clip(out, F(0.0), F(1.0)) * F(255.0)).astype(np.uint8)
    return q[..., [2, 1, 0, 3]] if bgra else q


def png_read_rgba8(path):
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    p, idat, w, h = 8, b"", 0, 0
    while p < len(b):
        ln, typ = struct.unpack(">I4s", b[p:p + 8])
        data = b[p + 8:p + 8 + ln]
        assert zlib.crc32(typ + data) == struct.unpack(">I", b[p + 8 + ln:p + 12 + ln])[0], "chunk CRC"
        if typ == b"IHDR":
            w, h, depth, ctype, comp, flt, inter = struct.unpack(">IIBBBBB", data)
            assert (depth, ctype, comp, flt, inter) == (8, 6, 0, 0, 0)
        elif typ == b"IDAT":
            idat += data
        p += 12 + ln
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, w * 4 + 1)
    out = np.zeros((h, w * 4), dtype=np.uint8)
    for y in range(h):
        f, line = raw[y, 0], raw[y, 1:].astype(np.int32)
        prev = out[y - 1].astype(np.int32) if y else np.zeros(w * 4, np.int32)
        if f == 0:
            out[y] = line
        elif f == 2:
            out[y] = (line + prev) & 255
        else:  # Sub / Average / Paeth need the running left neighbour
            cur = np.zeros(w * 4, np.int32)
            for i in range(w * 4):
                a = cur[i - 4] if i >= 4 else 0
                bb = prev[i]
                c = prev[i - 4] if i >= 4 else 0
                if f == 1:
                    pr = a
                elif f == 3:
                    pr = (a + bb) // 2
                else:
                    pa, pb, pc = abs(bb - c), abs(a - c), abs(a + bb - 2 * c)
                    pr = a if (pa <= pb and pa <= pc) else (bb if pb <= pc else c)
                cur[i] = (line[i] + pr) & 255
            out[y] = cur
    return out.reshape(h, w, 4)
