"""CPU tests of the callers' side of the hot path: c3dgs .npz reader, cameras.json scenes, PNG
write-out -- the library's native host code against the numpy oracle (oracle/ws_oracle_io.py) on generated files."""
import json
import os
import sys

import numpy as np
import pytest

from websplat import synth

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import ws_oracle_io as oio  # noqa: E402


def _quant_tuple(q):
    return {n: (getattr(q, n).zero_point, np.float32(getattr(q, n).scale)) for n in ("color_dc", "color_rest", "opacity", "scaling_factor")}


@pytest.mark.parametrize("compressed,with_sf,with_idx,sh_deg", [(True, True, True, 3), (False, True, True, 2),
                                                                (True, False, False, 1), (True, True, False, 0)])
def test_npz_reader_matches_oracle(ws, tmp_path, compressed, with_sf, with_idx, sh_deg):
    """io/npz.rs:59-225: DEFLATE and stored members, with and without the scaling-factor / codebook-index arrays."""
    a = synth.c3dgs_arrays(n=20_000, n_geometry=500, n_sh=333, seed=11 + sh_deg, sh_deg=sh_deg,
                           with_scaling_factor=with_sf, with_indices=with_idx)
    path = str(tmp_path / "scene.npz")
    synth.write_npz(path, a, compressed=compressed, kernel_size=np.array(0.1, dtype=np.float32),
                    mip_splatting=np.array(True), background_color=np.array([0.1, 0.2, 0.3], dtype=np.float32))
    got = ws.read_npz(path)
    ref = oio.npz_decode(path)
    assert got.compressed and got.num_points == ref["num_points"] == 20_000 and got.sh_deg == ref["sh_deg"] == sh_deg
    assert np.array_equal(got.gaussians, ref["gaussians"])          # GaussianCompressed records, byte-exact
    assert np.array_equal(got.sh_coefs, ref["sh"])                  # packed int8 SH records
    gc, rc = got.covars.view(np.uint16), ref["covars"].view(np.uint16)
    if with_sf:
        assert np.array_equal(gc, rc)                               # no libm call on this path: bit-exact f16
    else:
        # exp() comes from glibc on one side and numpy's SIMD kernel on the other (1 f32 ulp apart at most); the
        # off-diagonal terms cancel, so the bound is one f16 ulp of the LARGEST element of the matrix
        gf, rf = got.covars.view(np.float16).astype(np.float64), ref["covars"].view(np.float16).astype(np.float64)
        tol = np.abs(rf).max(axis=1, keepdims=True) * 2.0 ** -10
        assert (np.abs(gf - rf) <= tol).all() and (gc == rc).mean() > 0.99
    gq = _quant_tuple(got.quantization)
    for name, (zp, sc) in ref["quant"].items():
        assert gq[name][0] == zp and gq[name][1] == np.float32(sc), name
    assert got.kernel_size == pytest.approx(0.1) and got.mip_splatting is True
    assert np.allclose(got.background_color, [0.1, 0.2, 0.3])
    # new_compressed (io/mod.rs:107-150): bbox grown from the unit cube
    assert all(lo <= -1.0 for lo in got.aabb.min) and all(hi >= 1.0 for hi in got.aabb.max)


def test_npz_reader_scalar_dtypes_and_optionals(ws, tmp_path):
    """python floats / ints saved through np.savez arrive as float64 / int64; optional scalars may be absent."""
    a = synth.c3dgs_arrays(n=1000, n_geometry=50, n_sh=40, seed=5, sh_deg=1)
    a["opacity_scale"] = np.array(0.5)            # float64
    a["features_dc_zero_point"] = np.array(7)     # int64
    path = str(tmp_path / "s.npz")
    synth.write_npz(path, a, compressed=False)
    got = ws.read_npz(path)
    assert got.quantization.opacity.scale == 0.5 and got.quantization.color_dc.zero_point == 7
    assert got.kernel_size is None and got.mip_splatting is None and got.background_color is None


def test_npz_reader_errors(ws, tmp_path):
    a = synth.c3dgs_arrays(n=500, n_geometry=20, n_sh=20, seed=1, sh_deg=1)
    del a["rotation"]
    p1 = str(tmp_path / "missing.npz")
    synth.write_npz(p1, a)
    with pytest.raises(ws.WebSplatError, match="rotation missing"):
        ws.read_npz(p1)
    p2 = str(tmp_path / "garbage.npz")
    open(p2, "wb").write(b"PK\x03\x04" + b"\x00" * 100)
    with pytest.raises(ws.WebSplatError):
        ws.read_npz(p2)
    a = synth.c3dgs_arrays(n=500, n_geometry=20, n_sh=20, seed=1, sh_deg=1)
    p3 = str(tmp_path / "ok.npz")
    synth.write_npz(p3, a)
    raw = open(p3, "rb").read()
    p4 = str(tmp_path / "corrupt.npz")
    bad = bytearray(raw)
    bad[1000] ^= 0xFF  # inside the DEFLATE stream of xyz.npy: inflate or the CRC-32 must notice
    open(p4, "wb").write(bytes(bad))
    with pytest.raises(ws.WebSplatError):
        ws.read_npz(p4)
    with pytest.raises(ws.WebSplatError, match="cannot open"):
        ws.read_npz(str(tmp_path / "nope.npz"))


def _npy_member(header: str, payload: bytes = b"") -> bytes:
    """A version-1 .npy member with a hand-written header dict (what np.save would never emit)."""
    h = header.encode()
    pad = (64 - (10 + len(h) + 1) % 64) % 64
    h = h + b" " * pad + b"\n"
    return b"\x93NUMPY\x01\x00" + len(h).to_bytes(2, "little") + h + payload


def test_npz_reader_crafted_headers(ws, tmp_path):
    """Archives whose .npy headers or ZIP64 records overflow 64-bit size arithmetic must come back as errors, not as
    out-of-bounds reads or C++ exceptions across the ABI (each of these crashed or aborted an earlier build)."""
    import zipfile
    a = synth.c3dgs_arrays(n=300, n_geometry=20, n_sh=20, seed=2, sh_deg=1)
    base = str(tmp_path / "base.npz")
    synth.write_npz(base, a, compressed=False)

    def rewrite(name, member_bytes, out):
        with zipfile.ZipFile(base) as zin, zipfile.ZipFile(out, "w", zipfile.ZIP_STORED) as zout:
            for it in zin.infolist():
                zout.writestr(it.filename, member_bytes if it.filename == name + ".npy" else zin.read(it.filename))

    crafted = {
        # count() * item wraps to 0: the length check passed and elem_i32 read out of bounds
        "wrap_to_zero": ("gaussian_indices", _npy_member("{'descr': '<i8', 'fortran_order': False, 'shape': (2305843009213693952,), }")),
        # 'fortran_order': with no value -> std::out_of_range from compare()
        "fortran_novalue": ("xyz", _npy_member("{'descr': '<f2', 'shape': (300, 3), 'fortran_order':")),
        # hoff + hlen + bytes wraps, assign() threw length_error
        "huge_dim": ("opacity", _npy_member("{'descr': '|i1', 'fortran_order': False, 'shape': (18446744073709551615,), }")),
        "two_dims_overflow": ("scaling", _npy_member("{'descr': '|i1', 'fortran_order': False, 'shape': (4294967296, 4294967296), }")),
        "no_colon": ("rotation", _npy_member("{'descr' '|i1', 'shape' (20, 4)}")),
    }
    for tag, (name, blob) in crafted.items():
        out = str(tmp_path / f"{tag}.npz")
        rewrite(name, blob, out)
        with pytest.raises(ws.WebSplatError):
            ws.read_npz(out)

    # ZIP64 end record whose offsets wrap: locator -> z near 2^64, central directory offset + size wrapping
    raw = bytearray(open(base, "rb").read())
    eocd = raw.rfind(b"PK\x05\x06")
    for cd_off, cd_size, z in ((2**64 - 8, 16, None), (16, 2**64 - 8, None), (0, 0, 2**64 - 40)):
        z64 = (b"PK\x06\x06" + (44).to_bytes(8, "little") + b"\x2d\x00\x2d\x00" + bytes(8) + (1).to_bytes(8, "little") +
               (1).to_bytes(8, "little") + cd_size.to_bytes(8, "little") + cd_off.to_bytes(8, "little"))
        zpos = eocd if z is None else z
        loc = b"PK\x06\x07" + bytes(4) + zpos.to_bytes(8, "little") + (1).to_bytes(4, "little")
        tail = bytearray(raw[eocd:eocd + 22])
        tail[10:12] = b"\xff\xff"
        tail[12:16] = b"\xff\xff\xff\xff"
        tail[16:20] = b"\xff\xff\xff\xff"
        out = str(tmp_path / "z64.npz")
        open(out, "wb").write(bytes(raw[:eocd]) + z64 + loc + bytes(tail))
        with pytest.raises(ws.WebSplatError):
            ws.read_npz(out)
    # a member whose local header offset / compressed size point outside the file
    for field, val in ((42, 0xFFFFFFF0), (20, 0x7FFFFFFF)):
        bad = bytearray(raw)
        cd = bad.find(b"PK\x01\x02")
        bad[cd + field:cd + field + 4] = val.to_bytes(4, "little")
        out = str(tmp_path / "off.npz")
        open(out, "wb").write(bytes(bad))
        with pytest.raises(ws.WebSplatError):
            ws.read_npz(out)


def test_compressed_layout_follows_the_point_cloud(ws):
    """The int8 SH record stride is the POINT CLOUD's (3 * (deg + 1)^2 bytes); a degree-1 c3dgs cloud drawn by a
    default (degree-3) renderer must not be read with 48-byte records.  Host-side check only: prepare() on a renderer
    of lower degree is refused, of higher degree accepted (the GPU test renders it)."""
    # (the refusal itself needs a device; here: the descriptor the loader builds carries the cloud's own degree)
    a = synth.c3dgs_arrays(n=200, n_geometry=10, n_sh=10, seed=4, sh_deg=1)
    import tempfile, os
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "d1.npz")
        synth.write_npz(p, a)
        g = ws.read_npz(p)
    assert g.sh_deg == 1 and g.sh_coefs.size == 10 * 3 * 4


def _cams_json(n=19, seed=3):
    rng = np.random.default_rng(seed)
    cams = [c.to_json() for c in synth.orbit_cameras(n, 640, 480, 500.0, 510.0)]
    ids = rng.permutation(n) * 3 + 1                    # ids are neither dense nor in file order
    for c, i in zip(cams, ids):
        c["id"] = int(i)
        c["img_name"] = f"img_{int(i):04d}"
    cams[5]["id"] = cams[2]["id"]                       # duplicate id: the later entry wins (scene.rs:126-133)
    cams[5]["width"] = 999
    return json.dumps(cams)


def test_scene_from_json_matches_oracle(ws):
    """scene.rs:113-194: split by FILE position (every 8th = test), cameras() sorted by id, extend, nearest_camera."""
    text = _cams_json()
    sc = ws.Scene.from_json_text(text)
    ref = oio.scene_from_json(text)
    try:
        assert sc.num_cameras() == len(ref["cameras"]) == 18
        assert np.float32(sc.extend()) == ref["extend"]
        for split in (None, "train", "test"):
            got, want = sc.cameras(split), oio.scene_cameras(ref, split)
            assert [c.id for c in got] == [c["id"] for c in want]
            assert [c.split for c in got] == [c["split"] for c in want]
            for g, w in zip(got, want):
                assert (g.img_name, g.width, g.height) == (w["img_name"], w["width"], w["height"])
                assert np.array_equal(np.float32(g.position), np.float32(w["position"]))
                assert np.array_equal(np.float32(g.rotation), np.float32(w["rotation"]))
                assert (np.float32(g.fx), np.float32(g.fy)) == (np.float32(w["fx"]), np.float32(w["fy"]))
        dup = json.loads(text)[2]["id"]
        assert sc.camera(dup).width == 999 and sc.camera(10_000) is None
        rng = np.random.default_rng(0)
        for _ in range(20):
            p = rng.uniform(-5, 5, size=3)
            for split in (None, "train", "test"):
                assert sc.nearest_camera(p, split) == oio.scene_nearest(ref, p, split)
        # SceneCamera -> PerspectiveCamera (scene.rs:85-108) keeps working on parsed cameras
        cam = sc.cameras("train")[0].to_perspective()
        assert cam.fovx > 0 and cam.znear == pytest.approx(0.01) and cam.zfar == pytest.approx(100.0)
    finally:
        sc.close()


@pytest.mark.parametrize("text", ["", "{}", "[{\"id\": 1}]", "[1, 2", "[{\"id\":0,\"img_name\":\"a\",\"width\":1,\"height\":1,"
                                  "\"position\":[0,0],\"rotation\":[[1,0,0],[0,1,0],[0,0,1]],\"fx\":1,\"fy\":1}]"])
def test_scene_json_errors(ws, text):
    with pytest.raises(ws.WebSplatError):
        ws.Scene.from_json_text(text)


def test_scene_file_and_empty(ws, tmp_path):
    p = str(tmp_path / "cameras.json")
    synth.write_cameras_json(p, synth.orbit_cameras(9, 320, 240, 300.0, 300.0))
    sc = ws.Scene.from_json(p)
    assert sc.num_cameras() == 9 and len(sc.cameras("test")) == 2 and len(sc.cameras("train")) == 7
    sc.close()
    empty = ws.Scene.from_json_text("[]")
    assert empty.num_cameras() == 0 and empty.extend() == 0.0 and empty.nearest_camera([0, 0, 0]) is None
    empty.close()


@pytest.mark.parametrize("shape", [(1, 1), (7, 13), (240, 320)])
def test_png_roundtrip(ws, tmp_path, shape):
    rng = np.random.default_rng(shape[0])
    img = rng.integers(0, 256, size=(shape[0], shape[1], 4), dtype=np.uint8)
    p = str(tmp_path / "x.png")
    ws.write_png(p, img)
    assert np.array_equal(oio.png_read_rgba8(p), img)


def test_oracle_io_closed_forms():
    """Pin the numpy restatement itself: identity quaternion -> diag(s^2); texture read-back truncates; the display
    composite is `src + bg * (1 - a)` rounded to nearest."""
    q = np.array([[1.0, 0, 0, 0]], dtype=np.float32)
    s = np.array([[0.5, 2.0, 3.0]], dtype=np.float32)
    assert np.allclose(oio.build_cov(q, s), [[0.25, 0, 0, 4.0, 0, 9.0]])
    q90 = np.array([[np.sqrt(0.5), 0, 0, np.sqrt(0.5)]], dtype=np.float32)   # 90 degrees about z: x <-> y
    assert np.allclose(oio.build_cov(q90, s), [[4.0, 0, 0, 0.25, 0, 9.0]], atol=1e-6)
    img = np.array([[[0.999, 1.5, -0.2, 0.5]]], dtype=np.float16)
    assert oio.download_texture_u8(img).tolist() == [[[254, 255, 0, 127]]]
    src = np.array([[[0.25, 0.0, 0.5, 0.5]]], dtype=np.float32)
    assert oio.display_composite(src, (1.0, 1.0, 0.0, 1.0)).tolist() == [[[191, 128, 128, 255]]]
    assert oio.display_composite(src, (1.0, 1.0, 0.0, 1.0), bgra=True).tolist() == [[[128, 128, 191, 255]]]


# ---- robustness of the host parsers: arbitrary bytes must come back as an error code, never crash -------------------
from hypothesis import given, settings, strategies as st, HealthCheck  # noqa: E402


@settings(max_examples=300, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(st.text(alphabet=st.sampled_from(list("[]{}\",:0123456789.eE+-truefalsn \n\\u\"idpostnwhgfxyrm_")), max_size=200))
def test_scene_json_fuzz_never_crashes(ws, text):
    """RFC 8259 reader of cameras.json (scene.rs:113-135 via serde_json in the reference): any input either parses
    into a Scene or returns WS_ERR_*; the error travels as an exception, the process survives."""
    try:
        sc = ws.Scene.from_json_text(text)
    except ws.WebSplatError:
        return
    assert sc.num_cameras() >= 0
    sc.close()


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(st.data())
def test_scene_json_mutations_never_crash(ws, data):
    base = _cams_json(7)
    pos = data.draw(st.integers(0, len(base) - 1))
    kind = data.draw(st.sampled_from(["truncate", "delete", "replace", "duplicate"]))
    if kind == "truncate":
        text = base[:pos]
    elif kind == "delete":
        text = base[:pos] + base[pos + data.draw(st.integers(1, 20)):]
    elif kind == "replace":
        text = base[:pos] + data.draw(st.sampled_from(list("[]{},:\"x9-e. "))) + base[pos + 1:]
    else:
        text = base[:pos] + base[pos:pos + 30] + base[pos:]
    try:
        sc = ws.Scene.from_json_text(text)
    except ws.WebSplatError:
        return
    assert 0 <= sc.num_cameras() <= 8
    sc.close()


@settings(max_examples=120, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(st.data())
def test_npz_reader_mutations_never_crash(ws, tmp_path_factory, data):
    """ZIP / ZIP64 / .npy / DEFLATE reader (io/npz.rs via the npyz + zip crates in the reference): a valid c3dgs file
    with one region truncated, zeroed or overwritten with random bytes is either still readable or rejected."""
    global _NPZ_BASE
    try:
        raw = _NPZ_BASE
    except NameError:
        a = synth.c3dgs_arrays(n=300, n_geometry=16, n_sh=16, seed=9, sh_deg=1)
        p = str(tmp_path_factory.mktemp("fuzz") / "base.npz")
        synth.write_npz(p, a)
        raw = _NPZ_BASE = open(p, "rb").read()
    kind = data.draw(st.sampled_from(["truncate", "zero", "random", "central_dir"]))
    buf = bytearray(raw)
    if kind == "truncate":
        buf = buf[:data.draw(st.integers(0, len(raw) - 1))]
    else:
        lo = data.draw(st.integers(0, len(raw) - 1)) if kind != "central_dir" else data.draw(st.integers(max(0, len(raw) - 1200), len(raw) - 1))
        n = data.draw(st.integers(1, 64))
        fill = bytes(n) if kind == "zero" else data.draw(st.binary(min_size=n, max_size=n))
        buf[lo:lo + n] = fill[:max(0, min(n, len(raw) - lo))]
    path = str(tmp_path_factory.mktemp("fuzz") / "m.npz")
    open(path, "wb").write(bytes(buf))
    try:
        got = ws.read_npz(path)
    except ws.WebSplatError:
        return
    assert got.num_points >= 0


# ---- INRIA .ply reader on the host (io/ply.rs:28-196, io/mod.rs:63-105) ------------------------------------------------
@pytest.mark.parametrize("sh_deg,big", [(3, False), (3, True), (2, False), (1, True), (0, False)])
def test_ply_reader_matches_oracle(ws, oracle, tmp_path, sh_deg, big):
    rows = synth.scene_c1(n=3000, seed=60 + sh_deg, sh_deg=sh_deg)
    p = str(tmp_path / "scene.ply")
    synth.write_ply(p, rows, sh_deg, comments=["mip=false", "kernel_size=0.125", "background_color=1,0.5,0.25"], big_endian=big)
    got = ws.read_ply(p)
    assert got.num_points == 3000 and got.sh_deg == sh_deg and not got.compressed
    g, s = oracle.ply_rows_convert(rows, sh_deg)
    assert np.array_equal(got.gaussians, g) and np.array_equal(got.sh_coefs, s)          # byte-exact blobs
    bbox, center, up = oracle.pointcloud_stats(g, 28, oracle.make_aabb([0, 0, 0], [0, 0, 0]))
    assert np.array_equal(np.float32(got.aabb.min), np.float32(list(bbox.min)))
    assert np.array_equal(np.float32(got.aabb.max), np.float32(list(bbox.max)))
    assert np.array_equal(np.float32(got.center), np.float32(center))
    assert (got.up is None) == (up is None) and (up is None or np.allclose(got.up, up))
    assert got.mip_splatting is False and got.kernel_size == 0.125 and np.allclose(got.background_color, [1, 0.5, 0.25])
    # the same file through the rows path of the Python binding
    ref = ws.GenericGaussianPointCloud.from_ply_rows(rows, sh_deg)
    assert np.array_equal(ref.gaussians, got.gaussians) and np.array_equal(ref.sh_coefs, got.sh_coefs)


def test_ply_reader_errors(ws, tmp_path):
    rows = synth.scene_c1(n=100, seed=3, sh_deg=3)
    p = str(tmp_path / "ok.ply")
    synth.write_ply(p, rows, 3)
    raw = open(p, "rb").read()
    head_end = raw.index(b"end_header\n") + len(b"end_header\n")

    def expect(data, match=None):
        q = str(tmp_path / "bad.ply")
        open(q, "wb").write(data)
        with pytest.raises(ws.WebSplatError, match=match):
            ws.read_ply(q)

    expect(b"plx\n" + raw[4:], "magic")
    expect(raw[:head_end - len(b"end_header\n")], "end_header")
    expect(raw[:-40], "truncated")
    expect(raw.replace(b"binary_little_endian", b"ascii"), "ascii")
    expect(raw.replace(b"element vertex 100", b"element vertex 400000000"), "truncated")      # header lies about N
    expect(raw.replace(b"element vertex 100", b"element vertex 4000000000"), "2\\^30")          # must not be truncated to u32
    expect(raw.replace(b"element vertex 100", b"element vertex 4294967297"), "2\\^30")          # 2^32 + 1 is not "1 vertex"
    expect(raw.replace(b"property float f_rest_44\n", b""), "sh degree|layout")
    expect(raw.replace(b"property float opacity", b"property double opacity"), "sh degree|layout")
    # zero vertices: the reference's reader loops zero times and succeeds (io/ply.rs:164-196); bbox zeroed, centre 0/0
    q0 = str(tmp_path / "empty.ply")
    open(q0, "wb").write(raw[:head_end].replace(b"element vertex 100", b"element vertex 0"))
    empty = ws.read_ply(q0)
    assert empty.num_points == 0 and np.isnan(np.asarray(empty.center)).all()
    with pytest.raises(ws.WebSplatError, match="cannot open"):
        ws.read_ply(str(tmp_path / "nope.ply"))
    q = str(tmp_path / "mip.ply")
    synth.write_ply(q, rows, 3, comments=["mip=maybe"])
    with pytest.raises(ws.WebSplatError, match="mip"):
        ws.read_ply(q)
    synth.write_ply(q, rows, 3, comments=["background_color=red"])       # only warned about in the reference
    assert ws.read_ply(q).background_color is None
    for junk in ("kernel_size=abc", "kernel_size=", "kernel_size= 0.3", "kernel_size=0.3x"):   # parse::<f32>()? fails
        synth.write_ply(q, rows, 3, comments=[junk])
        with pytest.raises(ws.WebSplatError, match="kernel_size"):
            ws.read_ply(q)
    synth.write_ply(q, rows, 3, comments=["kernel_size=0.25"])
    assert ws.read_ply(q).kernel_size == 0.25
    with pytest.raises(ValueError, match="rows of"):                      # row length must fit the SH degree
        ws.GenericGaussianPointCloud.from_ply_rows(rows[:, :50], 3)


@settings(max_examples=150, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(st.data())
def test_ply_reader_mutations_never_crash(ws, tmp_path_factory, data):
    global _PLY_BASE
    try:
        raw = _PLY_BASE
    except NameError:
        p = str(tmp_path_factory.mktemp("fuzz") / "base.ply")
        synth.write_ply(p, synth.scene_c1(n=50, seed=4, sh_deg=2), 2, comments=["mip=true", "kernel_size=0.3"])
        raw = _PLY_BASE = open(p, "rb").read()
    head_end = raw.index(b"end_header\n") + 11
    kind = data.draw(st.sampled_from(["truncate", "header_byte", "header_del", "body", "number"]))
    buf = bytearray(raw)
    if kind == "truncate":
        buf = buf[:data.draw(st.integers(0, len(raw) - 1))]
    elif kind == "header_byte":
        buf[data.draw(st.integers(0, head_end - 1))] = data.draw(st.integers(0, 255))
    elif kind == "header_del":
        lo = data.draw(st.integers(0, head_end - 1))
        del buf[lo:lo + data.draw(st.integers(1, 40))]
    elif kind == "body":
        lo = data.draw(st.integers(head_end, len(raw) - 1))
        buf[lo:lo + 8] = data.draw(st.binary(min_size=8, max_size=8))[:len(raw) - lo]
    else:
        buf = bytearray(bytes(buf).replace(b"element vertex 50", b"element vertex " + str(data.draw(st.integers(0, 2 ** 40))).encode()))
    path = str(tmp_path_factory.mktemp("fuzz") / "m.ply")
    open(path, "wb").write(bytes(buf))
    try:
        got = ws.read_ply(path)
    except ws.WebSplatError:
        return
    assert got.num_points >= 0 and got.gaussians.shape == (got.num_points, 28)
