"""Seeded synthetic scenes in the reference's on-disk formats.

No scene assets exist in this environment (the reference's .gitignore excludes them and there is no
network), so every workload is generated: INRIA-layout PLY vertex rows (io/ply.rs:54-88 order),
cameras.json entries (scene.rs:13-24), and c3dgs-style compressed blobs (pointcloud.rs:14-22, 61-63).
Pure numpy; no dependence on the library or on the oracle.
"""
import json
from dataclasses import dataclass

import numpy as np

PLY_ROW_LEN = {d: 3 + 3 + 3 * (d + 1) ** 2 + 1 + 3 + 4 for d in range(4)}


def _rows(xyz, f_dc, f_rest_cm, opacity_logit, log_scale, rot):
    """Assemble N x 62 rows: x,y,z, nx,ny,nz, f_dc_0..2, f_rest_0..44 (channel-major), opacity, scale_0..2, rot_0..3."""
    n = xyz.shape[0]
    rows = np.zeros((n, 3 + 3 + 3 + f_rest_cm.shape[1] + 1 + 3 + 4), dtype=np.float32)
    rows[:, 0:3] = xyz
    rows[:, 6:9] = f_dc
    k = 9 + f_rest_cm.shape[1]
    rows[:, 9:k] = f_rest_cm
    rows[:, k] = opacity_logit
    rows[:, k + 1:k + 4] = log_scale
    rows[:, k + 4:k + 8] = rot
    return rows


def _sh(rng, n, sh_deg):
    f_dc = rng.uniform(-1.5, 1.5, size=(n, 3)).astype(np.float32)
    nrest = 3 * ((sh_deg + 1) ** 2 - 1)
    f_rest = (rng.standard_normal(size=(n, nrest)) * 0.1).astype(np.float32)
    return f_dc, f_rest


def scene_c1(n=10_000, seed=0, sh_deg=3):
    """C1: uniform cube of small Gaussians."""
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(-1.0, 1.0, size=(n, 3)).astype(np.float32)
    log_scale = rng.uniform(np.log(0.005), np.log(0.05), size=(n, 3)).astype(np.float32)
    rot = rng.standard_normal(size=(n, 4)).astype(np.float32)
    opacity = rng.uniform(-2.0, 4.0, size=n).astype(np.float32)
    f_dc, f_rest = _sh(rng, n, sh_deg)
    return _rows(xyz, f_dc, f_rest, opacity, log_scale, rot)


def scene_c2(n=1_200_000, seed=1, sh_deg=3):
    """C2 'bonsai-like': 60 % object blob, 40 % room shell."""
    rng = np.random.default_rng(seed)
    n_obj = int(n * 0.6)
    n_shell = n - n_obj
    obj = rng.standard_normal(size=(n_obj, 3)) * 0.35
    d = rng.standard_normal(size=(n_shell, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    shell = d * rng.uniform(3.0, 6.0, size=(n_shell, 1))
    xyz = np.concatenate([obj, shell]).astype(np.float32)
    perm = rng.permutation(n)  # file order is not spatial order
    xyz = xyz[perm]
    log_scale = (np.log(0.012) + 0.7 * rng.standard_normal(size=(n, 3))).astype(np.float32)
    rot = rng.standard_normal(size=(n, 4)).astype(np.float32)
    opacity = (1.0 + 2.0 * rng.standard_normal(size=n)).astype(np.float32)
    f_dc, f_rest = _sh(rng, n, sh_deg)
    return _rows(xyz, f_dc, f_rest, opacity, log_scale, rot)


def scene_c3(n=5_000_000, seed=2, sh_deg=3):
    """C3: 5 M tiny Gaussians filling a cube, all inside the frustum (sort stress)."""
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(-4.0, 4.0, size=(n, 3)).astype(np.float32)
    log_scale = (np.log(0.004) + 0.5 * rng.standard_normal(size=(n, 3))).astype(np.float32)
    rot = rng.standard_normal(size=(n, 4)).astype(np.float32)
    opacity = (1.0 + 2.0 * rng.standard_normal(size=n)).astype(np.float32)
    f_dc, f_rest = _sh(rng, n, sh_deg)
    return _rows(xyz, f_dc, f_rest, opacity, log_scale, rot)


def scene_realistic(n=1_000_000, seed=5, sh_deg=3):
    """'realistic1m': the SIZE DISTRIBUTION of a trained indoor scene (round-4 verdict item 8), which the uniform clouds above do
    not have -- every rate and tolerance of rounds 1-4 came from splats that span 1-6 tiles.  A trained 3DGS scene
    (io/ply.rs:50-100 reads what the INRIA trainer writes) is mostly small surface splats plus
      * a heavy tail of BACKGROUND-sized splats (walls, ceiling, sky dome): here 1.5 % with 8-40x the median scale, so that at
        1920x1080 more than 1 % of the visible splats cover >= 32 tiles of 32x32 px,
      * NEEDLES and DISCS: 15 % with an anisotropy of 10-50 : 1 (long axis sqrt(a) x the base size, the two others 1 / sqrt(a)),
        15 % with one axis squashed 10-50x,
      * a bimodal opacity: 45 % nearly opaque (sigmoid ~ 0.95-0.999), 35 % in between, 20 % at the pruning threshold of the
        trainer (alpha ~ 1/255 ... 0.02: splats that contribute almost nothing but are listed, sorted and walked all the same).
    Geometry: an object cluster on a table, a floor, two walls (surface splats lie IN their surface: discs), and floaters."""
    rng = np.random.default_rng(seed)
    n_obj, n_floor, n_wall = int(n * 0.45), int(n * 0.2), int(n * 0.2)
    n_float = n - n_obj - n_floor - n_wall
    obj = rng.standard_normal(size=(n_obj, 3)) * np.array([0.45, 0.3, 0.45]) + np.array([0.0, 0.1, 0.0])
    floor = np.stack([rng.uniform(-5, 5, n_floor), np.full(n_floor, 0.9) + 0.01 * rng.standard_normal(n_floor),
                      rng.uniform(-5, 5, n_floor)], axis=1)           # (camera y points down: the floor is at +y)
    w1 = np.stack([np.full(n_wall // 2, -5.0) + 0.01 * rng.standard_normal(n_wall // 2), rng.uniform(-3, 0.9, n_wall // 2),
                   rng.uniform(-5, 5, n_wall // 2)], axis=1)
    nw2 = n_wall - n_wall // 2
    w2 = np.stack([rng.uniform(-5, 5, nw2), rng.uniform(-3, 0.9, nw2), np.full(nw2, 5.0) + 0.01 * rng.standard_normal(nw2)], axis=1)
    floaters = rng.uniform(-4.5, 4.5, size=(n_float, 3)) * np.array([1.0, 0.6, 1.0])
    xyz = np.concatenate([obj, floor, w1, w2, floaters])
    kind = np.concatenate([np.zeros(n_obj, int), np.ones(n_floor, int), np.full(n_wall // 2, 2), np.full(nw2, 3), np.full(n_float, 4)])
    perm = rng.permutation(n)                                          # file order is not spatial order
    xyz, kind = xyz[perm].astype(np.float32), kind[perm]
    base = np.log(0.010) + 0.65 * rng.standard_normal(size=(n, 1))     # isotropic part, log-normal
    log_scale = base + 0.25 * rng.standard_normal(size=(n, 3))
    u = rng.uniform(size=n)
    axis = rng.integers(0, 3, size=n)
    stretch = np.exp(rng.uniform(np.log(10.0), np.log(50.0), size=n))
    needle, disc = u < 0.15, (u >= 0.15) & (u < 0.30)
    # a needle of anisotropy a: its long axis sqrt(a) x the base, the two others 1 / sqrt(a) (thin AND long, as trained edges are)
    log_scale[needle] -= 0.5 * np.log(stretch[needle])[:, None]
    log_scale[needle, axis[needle]] += np.log(stretch[needle])
    log_scale[disc, axis[disc]] -= np.log(stretch[disc])
    big = rng.uniform(size=n) < 0.015                                  # the background tail
    log_scale[big] += np.log(rng.uniform(8.0, 40.0, size=(int(big.sum()), 1)))
    # surface splats are flat in their surface's normal (floor: y; walls: x, z), axis-aligned rotation for those
    rot = rng.standard_normal(size=(n, 4))
    for k, ax in ((1, 1), (2, 0), (3, 2)):
        m = kind == k
        rot[m] = np.array([1.0, 0.0, 0.0, 0.0]) + 0.05 * rng.standard_normal(size=(int(m.sum()), 4))
        log_scale[m, ax] = np.minimum(log_scale[m, ax], np.log(0.002))
    v = rng.uniform(size=n)
    opacity = np.where(v < 0.45, rng.normal(4.5, 1.0, size=n),
                       np.where(v < 0.80, rng.normal(0.5, 1.5, size=n), rng.uniform(-5.6, -3.9, size=n)))
    f_dc, f_rest = _sh(rng, n, sh_deg)
    return _rows(xyz, f_dc, f_rest, opacity.astype(np.float32), log_scale.astype(np.float32), rot.astype(np.float32))


@dataclass
class SceneCamera:
    """scene.rs:13-24; rotation = 3 rows of the camera-to-world rotation (3DGS cameras.json)."""
    id: int
    img_name: str
    width: int
    height: int
    position: list
    rotation: list
    fx: float
    fy: float

    def to_json(self):
        return {"id": self.id, "img_name": self.img_name, "width": self.width, "height": self.height,
                "position": self.position, "rotation": self.rotation, "fx": self.fx, "fy": self.fy}


def look_at_camera(cam_id, position, target, width, height, fx, fy, down=(0.0, 1.0, 0.0)):
    """Camera looking from `position` to `target`; camera axes x right, y down, z forward (COLMAP/3DGS)."""
    p = np.asarray(position, dtype=np.float64)
    f = np.asarray(target, dtype=np.float64) - p
    f /= np.linalg.norm(f)
    r = np.cross(np.asarray(down, dtype=np.float64), f)
    r /= np.linalg.norm(r)
    d = np.cross(f, r)
    c2w = np.stack([r, d, f], axis=1)  # columns = camera axes in world coordinates
    return SceneCamera(cam_id, f"{cam_id:05d}", int(width), int(height),
                       [float(x) for x in p.astype(np.float32)],
                       [[float(v) for v in row] for row in c2w.astype(np.float32)], float(fx), float(fy))


def camera_c1(width=800, height=600):
    return SceneCamera(0, "00000", width, height, [0.0, 0.0, -3.0],
                       [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]], 800.0, 800.0)


def orbit_cameras(count, width, height, fx, fy, radius=4.0, height_off=1.0):
    """C2 / C4: cameras equally spaced on an orbit, looking at the origin (world 'down' is +y)."""
    cams = []
    for i in range(count):
        a = 2.0 * np.pi * i / count
        pos = [radius * np.cos(a), -height_off, radius * np.sin(a)]
        cams.append(look_at_camera(i, pos, [0.0, 0.0, 0.0], width, height, fx, fy))
    return cams


def camera_c3(width=1920, height=1080):
    # cube [-4,4]^3 seen from z = -12: nearest face at distance 8, half-size 4 -> tan(half fov) >= 0.5
    # vertical is the tight direction: fy = (h/2)/0.52
    fy = (height / 2.0) / 0.52
    return SceneCamera(0, "00000", width, height, [0.0, 0.0, -12.0],
                       [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]], fy, fy)


def write_cameras_json(path, cams):
    with open(path, "w") as f:
        json.dump([c.to_json() for c in cams], f)


def write_ply(path, rows, sh_deg=3, comments=(), big_endian=False):
    """INRIA 3DGS binary PLY (property order hard-coded by io/ply.rs:54-88)."""
    n = rows.shape[0]
    ncoef = (sh_deg + 1) ** 2
    props = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + \
            [f"f_rest_{i}" for i in range(3 * (ncoef - 1))] + ["opacity"] + [f"scale_{i}" for i in range(3)] + \
            [f"rot_{i}" for i in range(4)]
    assert rows.shape[1] == len(props)
    fmt = "binary_big_endian" if big_endian else "binary_little_endian"
    header = ["ply", f"format {fmt} 1.0"] + [f"comment {c}" for c in comments] + [f"element vertex {n}"] + \
             [f"property float {p}" for p in props] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(np.ascontiguousarray(rows, dtype=">f4" if big_endian else "<f4").tobytes())


def compressed_blobs(n=100_000, n_geometry=4096, n_sh=4096, seed=3, sh_deg=3, extent=1.0):
    """c3dgs-style GPU blobs (what io/npz.rs:59-225 produces), generated directly:
    GaussianCompressed 24 B, covariance codebook 12 B, packed int8 SH, quantisation block values."""
    rng = np.random.default_rng(seed)
    ncoef = (sh_deg + 1) ** 2
    g = np.zeros(n, dtype=np.dtype([("xyz", "<f4", 3), ("opacity", "i1"), ("scale_factor", "i1"), ("pad", "u1", 2),
                                    ("geometry_idx", "<u4"), ("sh_idx", "<u4")]))
    assert g.dtype.itemsize == 24
    g["xyz"] = rng.uniform(-extent, extent, size=(n, 3)).astype(np.float16).astype(np.float32)  # xyz is f16 on disk
    g["opacity"] = rng.integers(-128, 128, size=n, dtype=np.int8)
    g["scale_factor"] = rng.integers(-128, 128, size=n, dtype=np.int8)
    g["geometry_idx"] = rng.integers(0, n_geometry, size=n, dtype=np.uint32)
    g["sh_idx"] = rng.integers(0, n_sh, size=n, dtype=np.uint32)
    # covariance codebook of NORMALISED shapes (io/npz.rs:116-130): R diag(s) diag(s) R^T with |s| = 1
    q = rng.standard_normal(size=(n_geometry, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    s = np.abs(rng.standard_normal(size=(n_geometry, 3))) + 0.05
    s /= np.linalg.norm(s, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
                  np.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], -1),
                  np.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)], 1)
    L = R * s[:, None, :]
    cov = L @ np.transpose(L, (0, 2, 1))
    covars = np.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]],
                      -1).astype(np.float16)
    sh = rng.integers(-128, 128, size=(n_sh, 3 * ncoef), dtype=np.int8)
    quant = {
        "color_dc": (3, 0.012), "color_rest": (-2, 0.003),
        "opacity": (-128, 1.0 / 255.0),           # -> opacity in [0, 1]
        "scaling_factor": (200, 0.02),            # exp((i8 - 200) * 0.02) spans [0.0014, 0.23]
    }
    return {"gaussians": g.view(np.uint8).reshape(n, 24), "covars": covars.view(np.uint8).reshape(n_geometry, 12),
            "sh": sh.view(np.uint8).reshape(-1), "quant": quant, "sh_deg": sh_deg, "num_points": n}


def c3dgs_arrays(n=100_000, n_geometry=4096, n_sh=4096, seed=3, sh_deg=3, extent=1.0, with_scaling_factor=True,
                 with_indices=True):
    """The arrays of a c3dgs .npz (field names / dtypes / shapes of io/npz.rs:61-158): SURVEY 8(d) config C5."""
    rng = np.random.default_rng(seed)
    ncoef = (sh_deg + 1) ** 2
    m = n_geometry if with_indices else n
    k = n_sh if with_indices else n
    a = {
        "xyz": rng.uniform(-extent, extent, size=(n, 3)).astype(np.float16),
        "opacity": rng.integers(-128, 128, size=(n, 1), dtype=np.int8),
        "scaling": rng.integers(-128, 128, size=(m, 3), dtype=np.int8),
        "rotation": rng.integers(-128, 128, size=(m, 4), dtype=np.int8),
        "features_dc": rng.integers(-128, 128, size=(k, 3), dtype=np.int8),
        "features_rest": rng.integers(-128, 128, size=(k, ncoef - 1, 3), dtype=np.int8),
        "opacity_scale": np.array(1.0 / 255.0, dtype=np.float32), "opacity_zero_point": np.array(-128, dtype=np.int32),
        "rotation_scale": np.array(1.0 / 127.0, dtype=np.float32), "rotation_zero_point": np.array(3, dtype=np.int32),
        "features_dc_scale": np.array(0.012, dtype=np.float32), "features_dc_zero_point": np.array(3, dtype=np.int32),
        "features_rest_scale": np.array(0.003, dtype=np.float32), "features_rest_zero_point": np.array(-2, dtype=np.int32),
    }
    # a few exactly-zero rotations would normalise to NaN in the reference as well; keep the codebook regular
    a["rotation"][np.all(a["rotation"] == 3, axis=1)] = np.array([100, 3, 3, 3], dtype=np.int8)
    if with_scaling_factor:
        a["scaling_scale"] = np.array(0.01, dtype=np.float32)
        a["scaling_zero_point"] = np.array(-130, dtype=np.int32)     # (i8 + 130) * 0.01 > 0: a direction to normalise
        a["scaling_factor"] = rng.integers(-128, 128, size=(n,), dtype=np.int8)
        a["scaling_factor_scale"] = np.array(0.02, dtype=np.float32)
        a["scaling_factor_zero_point"] = np.array(200, dtype=np.int32)
    else:
        a["scaling_scale"] = np.array(0.02, dtype=np.float32)       # exp((i8 - 100) * 0.02): absolute scales
        a["scaling_zero_point"] = np.array(100, dtype=np.int32)
    if with_indices:
        a["gaussian_indices"] = rng.integers(0, n_geometry, size=(n,), dtype=np.int32)
        a["feature_indices"] = rng.integers(0, n_sh, size=(n,), dtype=np.int32)
    return a


def write_npz(path, arrays, compressed=True, **extra):
    """np.savez_compressed (DEFLATE members, what c3dgs writes) or np.savez (stored members)."""
    (np.savez_compressed if compressed else np.savez)(path, **arrays, **extra)
