"""CPU tests of the product's host side (no GPU needed): the C-ABI library loads and exports every symbol
include/websplat.h declares; camera / uniform / loader-prep code agrees with the oracle's restatement of
camera.rs, renderer.rs:321-343,620-651, io/ply.rs:50-100, io/mod.rs:63-105."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from websplat import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol(ws):
    header = open(os.path.join(ROOT, "include", "websplat.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(ws_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 40
    from websplat import _lib
    raw = C.CDLL(_lib.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(raw, s)]
    assert not missing, f"declared in websplat.h but not exported: {missing}"
    unbound = sorted(declared - set(_lib.SIGNATURES))
    assert not unbound, f"declared in websplat.h but not bound by the Python stub: {unbound}"
    assert ws.lib.ws_abi_version() == 3


def test_the_library_reads_no_environment_variable(ws):
    """Round 6: a drop-in library must not be steered by the environment of whoever loads it.  Every switch travels in
    ws_context_config; the translation of the WS_* variables lives in the harness (websplat.config_from_env,
    include/websplat_env.h).  The product library does not even import getenv, and no WS_* name is in its strings."""
    import subprocess
    from websplat import _lib
    und = subprocess.run(["nm", "-D", "-u", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in und, "the library imports getenv"
    blob = open(_lib.LIB_PATH, "rb").read()
    for name in (b"WS_GRAPH", b"WS_BLEND_ORDER", b"WS_TILE_SHAPE", b"WS_DEPTH_SORT\0", b"WS_CAPTURE", b"WS_BATCH_QUEUE_DEPTH"):
        assert name not in blob, name
    c = ws.config_from_env({"WS_BLEND_ORDER": "1", "WS_TILE_SHAPE": "4x2", "WS_BIN_SHIFT": "0", "WS_DEPTH_DIGIT_BITS": "9",
                            "WS_DEPTH_SORT": "onesweep", "WS_PLY_DECODE": "host"}, debug_cut=2)
    assert (c.struct_size, c.blend_order, c.tile_qw, c.tile_qh, c.bin_request, c.depth_digit_bits, c.exp_depth_sort,
            c.ply_decode_host, c.debug_cut, c.batch_queue_depth, c.exp_batch_k1) == (128, 1, 4, 2, 0, 9, 1, 1, 2, -1, 1)
    d = ws.config_from_env({})
    assert (d.use_graph, d.depth_skip_top, d.blend_order, d.blend_split, d.bin_request, d.batch_threads, d.tile_qw, d.tile_qh,
            d.depth_digit_bits, d.exp_depth_sort) == (0, 1, -1, -1, 1, -1, 4, 4, 0, 0)


def test_no_gpu_means_loud_failure(ws):
    """No CPU fallback: without a device, context creation must fail with a HIP error, not emulate."""
    try:
        c = ws.Context(0)
    except ws.WebSplatError as e:
        assert "no HIP device" in str(e) and e.code == -2
    else:
        c.close()
        pytest.skip("GPU present")


def _uniform_arrays(u):
    return {k: np.array(list(getattr(u, k)), dtype=np.float32) for k in ("view", "view_inv", "proj", "proj_inv", "viewport", "focal")}


@pytest.mark.parametrize("cam_idx", [0, 5, 11])
def test_camera_uniform_matches_oracle(ws, oracle, cam_idx):
    cj = synth.orbit_cameras(16, 1200, 799, 1200.0, 1150.0)[cam_idx]
    cam = ws.PerspectiveCamera.from_scene_camera(cj.position, cj.rotation, cj.fx, cj.fy, cj.width, cj.height)
    ocam = oracle.scene_camera_to_perspective(cj.position, cj.rotation, cj.fx, cj.fy, cj.width, cj.height)
    assert np.array_equal(np.array(cam.rotation, np.float32), np.array(list(ocam.rotation), np.float32))
    assert (cam.fovx, cam.fovy, cam.fov2view_ratio) == (ocam.fovx, ocam.fovy, ocam.fov2view_ratio)
    bbox = ws.Aabb([-6, -6, -6], [6, 6, 6])
    cam.fit_near_far(bbox)
    oracle.fit_near_far(ocam, oracle.make_aabb(bbox.min, bbox.max))
    assert (cam.znear, cam.zfar) == (ocam.znear, ocam.zfar)
    u = _uniform_arrays(cam.uniform((1200, 799)))
    o = _uniform_arrays(oracle.camera_uniform(ocam, 1200, 799))
    for k in u:  # same f32 operation sequence on both sides -> bit-identical
        assert np.array_equal(u[k].view(np.uint32), o[k].view(np.uint32)), k


def test_negative_determinant_rotation_is_flipped(ws, oracle):
    rot = [[1.0, 0.0, 0.0], [0.0, -1.0, 0.0], [0.0, 0.0, 1.0]]  # det = -1 (scene.rs:90-96)
    cam = ws.PerspectiveCamera.from_scene_camera([0, 0, -3], rot, 500, 500, 640, 480)
    ocam = oracle.scene_camera_to_perspective([0, 0, -3], rot, 500, 500, 640, 480)
    assert np.allclose(cam.rotation, list(ocam.rotation))
    assert np.allclose(cam.rotation, [1, 0, 0, 0])


def test_ply_rows_convert_matches_oracle(ws, oracle):
    rows = synth.scene_c1(n=5000, seed=11)
    rows[0, 54] = 40.0    # sigmoid saturation
    rows[1, 54] = -40.0
    rows[2, 55:58] = [3.0, -9.0, 0.0]  # extreme scales (f16 overflow / underflow of the covariance)
    pc = ws.GenericGaussianPointCloud.from_ply_rows(rows, 3)
    g, sh = oracle.ply_rows_convert(rows, 3)
    assert np.array_equal(pc.gaussians, g)
    assert np.array_equal(pc.sh_coefs, sh)
    # spot-check the record layout (pointcloud.rs:38-45, io/mod.rs:65)
    assert np.array_equal(g[7, :12].view(np.float32), rows[7, :3])
    op = 1.0 / (1.0 + np.exp(-np.float64(rows[7, 54])))
    assert np.isclose(g[7, 12:14].view(np.float16)[0], op, atol=1e-3)
    assert g[7, 14] == 0 and g[7, 15] == 0
    sh16 = sh[7].view(np.float16).reshape(16, 3)
    assert np.allclose(sh16[0], rows[7, 6:9], atol=2e-3)
    rest = rows[7, 9:54].reshape(3, 15)  # channel-major in the file
    assert np.allclose(sh16[1:], rest.T, atol=1e-3)
    bbox, center, up = oracle.pointcloud_stats(g, 28, oracle.make_aabb([0, 0, 0], [0, 0, 0]))
    assert np.array_equal(np.array(pc.aabb.min, np.float32), np.array(list(bbox.min), np.float32))
    assert np.array_equal(np.array(pc.aabb.max, np.float32), np.array(list(bbox.max), np.float32))
    assert np.array_equal(np.array(pc.center, np.float32), np.array(center, np.float32))
    assert pc.up is None and up is None  # bbox radius < 10 -> up dropped (io/mod.rs:88-90)


@pytest.mark.parametrize("sh_deg", [0, 1, 2])
def test_ply_rows_convert_lower_degrees(ws, oracle, sh_deg):
    full = synth.scene_c1(n=300, seed=3, sh_deg=sh_deg)
    assert full.shape[1] == synth.PLY_ROW_LEN[sh_deg]
    pc = ws.GenericGaussianPointCloud.from_ply_rows(full, sh_deg)
    g, sh = oracle.ply_rows_convert(full, sh_deg)
    assert np.array_equal(pc.gaussians, g) and np.array_equal(pc.sh_coefs, sh)
    ncoef = (sh_deg + 1) ** 2
    assert not sh.view(np.float16).reshape(-1, 16, 3)[:, ncoef:].any()  # unused coefficients are zero


def test_up_vector_large_scene(ws, oracle):
    rng = np.random.default_rng(2)
    rows = synth.scene_c1(n=4000, seed=2)
    rows[:, 0] = rng.uniform(-30, 30, 4000)   # wide slab in x/z, thin in y -> plane normal ~ +-y
    rows[:, 2] = rng.uniform(-30, 30, 4000)
    rows[:, 1] = rng.uniform(-0.2, 0.2, 4000)
    pc = ws.GenericGaussianPointCloud.from_ply_rows(rows, 3)
    g, _ = oracle.ply_rows_convert(rows, 3)
    _, _, up = oracle.pointcloud_stats(g, 28, oracle.make_aabb([0, 0, 0], [0, 0, 0]))
    assert pc.up is not None and up is not None
    assert np.allclose(pc.up, up, atol=1e-6)
    assert pc.up[1] > 0.99


def test_write_ply_layout(tmp_path):
    rows = synth.scene_c1(n=10, seed=1)
    p = tmp_path / "a.ply"
    synth.write_ply(str(p), rows, 3, comments=["mip=true", "kernel_size=0.1"])
    raw = p.read_bytes()
    head, body = raw.split(b"end_header\n", 1)
    assert head.startswith(b"ply\nformat binary_little_endian 1.0\n")
    assert head.count(b"property float") == 62 and len(body) == 10 * 248


# ---- the compositing pass's staging step (blend_stage.h), host twin through the ABI -----------------------------
def _random_splat_words(rng, viewport, tile_origin, tile):
    """A 20-B Splat record (pointcloud.rs:352-358) whose ellipse lies somewhere around the tile: sizes from
    sub-pixel to thousands of pixels, anisotropy up to ~1e4."""
    W, H = viewport
    s1 = np.exp(rng.uniform(np.log(0.3), np.log(3000.0)))
    s2 = s1 * np.exp(rng.uniform(-1, 1)) if rng.random() < 0.4 else np.exp(rng.uniform(np.log(0.3), np.log(3000.0)))
    th = rng.uniform(0, 2 * np.pi)
    reach = 2.3 * max(s1, s2) + max(tile)
    cx = tile_origin[0] + tile[0] / 2 + rng.uniform(-1, 1) * reach
    cy = tile_origin[1] + tile[1] / 2 + rng.uniform(-1, 1) * reach
    if rng.random() < 0.25:
        cx = tile_origin[0] + rng.uniform(0, tile[0])
        cy = tile_origin[1] + rng.uniform(0, tile[1])
    c, s = np.cos(th), np.sin(th)
    # M = [[m00, m01], [m10, m11]] in pixels; v1 = (m00 / W, -m10 / H), v2 = (m01 / W, -m11 / H)
    m00, m01, m10, m11 = c * s1, -s * s2, s * s1, c * s2
    halves = np.array([m00 / W, -m10 / H, m01 / W, -m11 / H, cx / W * 2 - 1, 1 - cy / H * 2,
                       rng.random(), rng.random(), rng.random(), rng.random()], dtype=np.float32).astype(np.float16)
    return halves.view(np.uint32)


@pytest.mark.parametrize("tile", [(16, 16), (32, 16), (32, 32)])
def test_quadrant_mask_never_drops_a_covered_pixel(ws, tile):
    """The staged record's quadrant mask only prunes work: every pixel centre the per-pixel test keeps
    (a' <= 2*CUTOFF*log2 e, evaluated in f32 exactly as the kernel does) must lie in a quadrant whose bit is set;
    and the mask must stay tight (few quadrants flagged that hold no kept pixel)."""
    rng = np.random.default_rng(7)
    viewport = (1920.0, 1080.0)
    cut = np.float32(2 * 2.3539888583335364 * 1.4426950408889634)
    qw = tile[0] // 8
    ys, xs = np.mgrid[0:tile[1], 0:tile[0]]
    lx = (xs + 0.5).astype(np.float32)
    ly = (ys + 0.5).astype(np.float32)
    quad = (ys // 8) * qw + (xs // 8)
    needed = flagged = 0
    for trial in range(6000):
        origin = (float(tile[0] * rng.integers(0, 40)), float(tile[1] * rng.integers(0, 30)))
        words = _random_splat_words(rng, viewport, origin, tile)
        rec, mask = ws.stage_splat(words, viewport, origin, tile)
        if not np.all(np.isfinite(rec[:6])):
            continue
        # a' per pixel centre: same operation order as blend_composite (fma chains), in f64-free f32
        p0 = rec[0] * lx + (rec[1] * ly + rec[2])
        p1 = rec[3] * lx + (rec[4] * ly + rec[5])
        a = p0 * p0 + p1 * p1
        # allow for fma-vs-separate rounding: treat anything within a relative 1e-5 of the cut-off as kept
        kept = a <= cut * np.float32(1.00001)
        truth = 0
        for q in np.unique(quad[kept]):
            truth |= 1 << int(q)
        assert truth & ~mask == 0, (trial, hex(truth), hex(mask), words)
        needed += bin(truth).count("1")
        flagged += bin(mask).count("1")
    assert needed > 2000
    assert flagged <= needed * 1.02 + 10


# ---- the binning footprint (footprint.h), host twin through the ABI ---------------------------------------------
def _kept_tiles_brute_force(words, viewport, tile):
    """Tiles holding at least one pixel centre the blend's per-pixel test keeps: the staging step's tile-local affine
    form (blend_stage.h decode) evaluated in f32 for every pixel of the viewport, with the relative 1e-5 allowance for
    fma-vs-separate rounding the quadrant-mask test uses."""
    f = np.float32
    W, H = f(viewport[0]), f(viewport[1])
    h = np.asarray(words[:3], dtype=np.uint32).view(np.float16).astype(np.float32)
    m00, m01 = h[0] * W, h[2] * W
    m10, m11 = -h[1] * H, -h[3] * H
    det = f(m00 * m11) - f(m01 * m10)
    if not np.isfinite(det) or det == 0:
        return None
    with np.errstate(all="ignore"):
        inv = f(1.2011224087864498) / det
        i00, i01, i10, i11 = m11 * inv, -m01 * inv, -m10 * inv, m00 * inv
        if not np.all(np.isfinite([i00, i01, i10, i11])):
            return None
        tw, th = tile
        ntx = -(-int(viewport[0]) // tw)
        ys, xs = np.mgrid[0:int(viewport[1]), 0:int(viewport[0])]
        ox, oy = (xs // tw * tw).astype(np.float32), (ys // th * th).astype(np.float32)
        cxl = (h[4] * f(0.5) + f(0.5)) * W - ox
        cyl = (f(0.5) - h[5] * f(0.5)) * H - oy
        c0 = -(i00 * cxl + i01 * cyl)
        c1 = -(i10 * cxl + i11 * cyl)
        lx, ly = (xs - ox + f(0.5)).astype(np.float32), (ys - oy + f(0.5)).astype(np.float32)
        p0 = i00 * lx + (i01 * ly + c0)
        p1 = i10 * lx + (i11 * ly + c1)
        a = p0 * p0 + p1 * p1
    cut = f(2 * 2.3539888583335364 * 1.4426950408889634)
    kept = a <= cut * f(1.00001)
    return set(np.unique((ys // th * ntx + xs // tw)[kept]).tolist())


@pytest.mark.parametrize("tile", [(32, 32), (32, 16), (16, 16)])
def test_footprint_never_drops_a_touched_tile(ws, tile):
    """Binning by the kept ellipse (footprint.h): the tile list of a splat must contain every tile in which the blend's
    per-pixel test keeps a pixel centre, each tile once, in row-major order; and it must stay tight -- the point of the
    exercise is NOT to list the bounding rectangle."""
    rng = np.random.default_rng(11)
    viewport = (416, 304)   # 13 x 9.5 tiles of 32: a ragged last row
    ntx = -(-viewport[0] // tile[0])
    needed = listed = rect = 0
    trials = 0
    for trial in range(2500):
        origin = (float(rng.integers(0, viewport[0])), float(rng.integers(0, viewport[1])))
        words = _random_splat_words(rng, tuple(float(v) for v in viewport), origin, (1, 1))
        if trial % 7 == 0:   # long thin needles through the viewport: the case a bounding rectangle is worst at
            hv = words.view(np.float16).copy()
            s1, s2, th_ = rng.uniform(100, 400), rng.uniform(0.4, 3.0), rng.uniform(0, 2 * np.pi)
            c, s = np.cos(th_), np.sin(th_)
            hv[0:4] = np.array([c * s1 / viewport[0], -s * s1 / viewport[1], -s * s2 / viewport[0], -c * s2 / viewport[1]],
                               dtype=np.float16)
            words = hv.view(np.uint32)
        truth = _kept_tiles_brute_force(words, viewport, tile)
        if truth is None:
            continue
        tiles = ws.footprint_tiles(words, viewport, tile)
        assert len(set(tiles.tolist())) == len(tiles), (trial, tiles)            # each tile once
        assert np.all(np.diff(tiles.astype(np.int64)) > 0), (trial, tiles)       # rows top to bottom, columns left to right
        assert truth <= set(tiles.tolist()), (trial, sorted(truth - set(tiles.tolist())), words)
        trials += 1
        needed += len(truth)
        listed += len(tiles)
        if len(tiles):
            tx, ty = tiles % ntx, tiles // ntx
            rect += int((tx.max() - tx.min() + 1) * (ty.max() - ty.min() + 1))
    assert trials > 2000 and needed > 10000
    assert listed <= needed * 1.03 + 20, (listed, needed)    # tight: within 3 % of the tiles that hold a kept pixel
    assert rect >= listed * 1.1                               # ... and visibly fewer than the bounding rectangles


def test_openmp_workers_sleep_after_a_host_region(ws):
    """The library's host loops are OpenMP regions; LLVM's runtime keeps a finished region's workers spinning for 200 ms
    (KMP_BLOCKTIME) -- on the GPU box 128 busy threads next to the thread that enqueues frames, measured as 0.2 s at 2 600
    instead of 16 800 frames/s after every ws_pointcloud_create (profiles/r03/slowmode_openmp_blocktime_v23.txt).  The
    regions hold an OmpQuietWorkers guard: right after a conversion the process must be idle."""
    import os
    import time
    if (os.cpu_count() or 1) < 2 or os.environ.get("KMP_BLOCKTIME") or os.environ.get("OMP_WAIT_POLICY"):
        pytest.skip("needs several cores and the OpenMP runtime's default wait policy")
    rows = synth.scene_c1(n=60_000, seed=12)
    ws.GenericGaussianPointCloud.from_ply_rows(rows, 3)   # the OpenMP runtime and its team exist from here on
    burn = []
    for _ in range(3):
        ws.GenericGaussianPointCloud.from_ply_rows(rows, 3)
        c0 = time.process_time()      # CPU time of ALL threads of the process
        time.sleep(0.1)
        burn.append(time.process_time() - c0)
    # spinning workers would burn ~0.1 s per core (0.7 s on 8 cores); a sleeping team costs next to nothing
    assert min(burn) < 0.05, burn


def test_packed_rectangle_at_both_binning_sizes(ws):
    """The 4-byte word a splat carries through the depth sort is its tile rectangle; K1 sums its tile count at the compositing
    tile and at 2 x 2 of them, the binning kernels read it in either unit (rect_tiles / rect_tiles64 / rect_coarse,
    csrc/ws_internal.h).  Against a brute-force walk over the tiles."""
    rng = np.random.default_rng(71)
    cases = [(0, 0, 1, 1), (255, 255, 1, 1), (0, 0, 256, 256), (1, 1, 1, 1), (1, 0, 2, 3), (254, 3, 2, 2)]
    for _ in range(3000):
        x0, y0 = int(rng.integers(0, 256)), int(rng.integers(0, 256))
        cases.append((x0, y0, int(rng.integers(1, 257 - x0)), int(rng.integers(1, 257 - y0))))
    for x0, y0, w, h in cases:
        rect = x0 | (y0 << 8) | ((w - 1) << 16) | ((h - 1) << 24)
        if rect == 0xFFFFFFFF:   # the one rectangle whose word is the "no tile" marker: never produced (x0 = 255 has w = 1)
            continue
        tiles, coarse, rc = ws.packed_rect(rect)
        cells = {(x >> 1, y >> 1) for x in range(x0, x0 + w) for y in range(y0, y0 + h)}
        assert tiles == w * h and coarse == len(cells), (x0, y0, w, h)
        cx0, cy0, cw, ch = rc & 0xFF, (rc >> 8) & 0xFF, ((rc >> 16) & 0xFF) + 1, (rc >> 24) + 1
        assert {(x, y) for x in range(cx0, cx0 + cw) for y in range(cy0, cy0 + ch)} == cells
    assert ws.packed_rect(0xFFFFFFFF) == (0, 0, 0xFFFFFFFF)


def test_binning_decision_from_k1_sums(ws):
    """bin_shift_decide: 2 x 2 binning iff the frame asks the device to decide and the summed tile counts shrink by 1.5x or
    more; the sixteen slots K1's workgroups add into are folded first; never / always override."""
    assert ws.binning_decision(1, [150], [100]) == 1
    assert ws.binning_decision(1, [149], [100]) == 0
    assert ws.binning_decision(1, [10] * 16, [7] * 16) == 0          # 160 : 112 = 1.43
    assert ws.binning_decision(1, [10] * 15 + [18], [7] * 16) == 1   # 168 : 112 = 1.50
    assert ws.binning_decision(1, [], []) == 0 and ws.binning_decision(1, [0], [0]) == 0   # an empty frame
    assert ws.binning_decision(0, [1000], [100]) == 0 and ws.binning_decision(2, [100], [100]) == 1
    assert ws.binning_decision(1, [4_000_000_000 // 16] * 16, [2_000_000_000 // 16] * 16) == 1   # sums near 2^32


def test_context_config_layout_matches_the_c_header(ws, tmp_path):
    """The ctypes mirror of ws_context_config (and the field order INTEGRATION.md's Rust stub lists) against the C header itself:
    a C99 program prints sizeof and every field's offset; the Python struct must agree field for field, and
    ws_context_config_init must fill the documented defaults."""
    import subprocess
    from websplat import _lib
    fields = [n for n, _ in _lib.ws_context_config._fields_]
    src = ["#include <stdio.h>", "#include <stddef.h>", '#include "websplat.h"', '#include "websplat_env.h"', "int main(void) {",
           '  printf("sizeof %zu\\n", sizeof(ws_context_config));']
    src += [f'  printf("{n} %zu\\n", offsetof(ws_context_config, {n}));' for n in fields]
    src += ["  return 0;", "}"]
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    assert int(out["sizeof"]) == C.sizeof(_lib.ws_context_config) == 128
    for n in fields:
        assert int(out[n]) == getattr(_lib.ws_context_config, n).offset, n
    cfg = _lib.ws_context_config()
    ws.lib.ws_context_config_init(C.byref(cfg))
    got = {n: (list(getattr(cfg, n)) if n == "reserved" else getattr(cfg, n)) for n in fields}
    want = dict.fromkeys(fields, 0)
    want.update(struct_size=128, depth_skip_top=1, blend_order=-1, blend_split=-1, bin_request=1, batch_threads=-1, batch_queue_depth=-1,
                blend_tpw_log2=-1, tile_qw=4, tile_qh=4, exp_batch_k1=1, blend_async=-1, reserved=[0] * 6)
    assert got == want


def test_depth_sort_range_decision(ws):
    """ws_internal.h depth_range_decide through its host twin: base = min(key) with the first digit's bits cleared (the first pass runs
    before anybody knows the base), the fourth pass is the identity iff max - base < radix^3 (2^24 at 8 bits, 2^27 at 9), the span class
    the next frame's digit width follows is taken on the 8-bit base whatever the radix, no key -> nothing decided, and a frame that
    holds a key of 0xFFFFFFFF keeps base 0 (the scatter kernels exempt that value -- also their padding key -- from the subtraction,
    the histogram kernels do not: ADVICE r05)."""
    def decide(lo, hi, digits, have=1):
        b, s, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        assert ws.lib.ws_debug_depth_range(lo, hi, have, digits, C.byref(b), C.byref(s), C.byref(c)) == 0
        return b.value, s.value, c.value
    assert decide(0, 0, 256, have=0) == (0, 0, 0) and decide(0, 0, 512, have=0) == (0, 0, 0)
    lo = 0x40801234
    assert decide(lo, lo + (1 << 24) - 0x35, 256) == (0x40801200, 1, 1)          # span just under 2^24 above the 256-aligned base
    assert decide(lo, lo + (1 << 24), 256) == (0x40801200, 0, 2)                  # ... one step over: four 8-bit passes
    assert decide(lo, lo + (1 << 24), 512) == (0x40801200, 1, 2)                  # nine-bit digits: three passes, class says "not < 2^24"
    assert decide(lo, 0x40801000 + (1 << 27) - 1, 512) == (0x40801200 & ~0x1FF, 1, 2)
    assert decide(lo, (0x40801200 & ~0x1FF) + (1 << 27), 512)[1] == 0            # beyond 2^27: the fourth 9-bit pass runs
    assert decide(5, 5, 256) == (0, 1, 1) and decide(0x1FF, 0x200, 512) == (0, 1, 1)
    assert decide(0xFFFFFF00, 0xFFFFFFFF, 256) == (0, 0, 2)                       # a real key of 0xFFFFFFFF: no base
    assert decide(0xFFFFFF00, 0xFFFFFFFE, 256) == (0xFFFFFF00, 1, 1)
    with pytest.raises(ws.WebSplatError):
        ws.check(ws.lib.ws_debug_depth_range(1, 2, 1, 300, C.byref(C.c_uint32()), C.byref(C.c_uint32()), C.byref(C.c_uint32())))
