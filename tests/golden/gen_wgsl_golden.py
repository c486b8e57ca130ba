#!/usr/bin/env python3
"""Generates tests/golden/wgsl_*.npz: outputs of the reference's OWN shader source, executed by oracle/wgsl_exec.py.

Run in the build container (the reference checkout is not on the GPU box and nothing at test time reads it):

    python tests/golden/gen_wgsl_golden.py            # all cases
    python tests/golden/gen_wgsl_golden.py k1_default # one case

The shader text is read from $WEBSPLAT_REFERENCE/src/shaders (default /root/reference) at generation time and is NOT
stored: the fixtures hold only the seeded input buffers (the bytes a wgpu binding would see) and the output buffers the
shader wrote.  Bindings and the MAX_SH_DEG prefix follow renderer.rs:379-392 (build_shader) and :394-420 (bind groups).

  wgsl_k1_<case>.npz    preprocess.wgsl `preprocess`            -> points_2d, sort_depths, keys_size, dispatch_x (+ the
                        invocation that drew each store index)
  wgsl_k1c_<case>.npz   preprocess_compressed.wgsl `preprocess` -> the same
  wgsl_sort_<case>.npz  radix_sort.wgsl driven as GPURSSorter::record_sort drives it -> sorted keys and payload
  wgsl_frame.npz        K1 + stable sort + the instanced draw with the pipeline's blend state: a whole 320x240 frame
  wgsl_k6_fragments.npz gaussian.wgsl `vs_main` (4 vertices per instance) and `fs_main` at pixel centres; the
                        screen_pos a fragment receives is the rasteriser's linear interpolation of the four vertices'
                        values, restated here in float64 and rounded to f32 (the one step no shader text covers).

Inputs come from the repository's synthetic scene generators and the oracle's host math (uniform structs as bytes);
the interpreter decodes them with WGSL's own layout rules, so a layout disagreement shows up as a value mismatch.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "web-splat_amd")]
import wgsl_exec as W  # noqa: E402

REF = os.environ.get("WEBSPLAT_REFERENCE", "/root/reference")


def shader(name, max_sh_deg=None):
    src = open(os.path.join(REF, "src", "shaders", name)).read()
    if max_sh_deg is not None:  # renderer.rs:379-392
        src = "\n        const MAX_SH_DEG:u32 = %du;\n        %s" % (max_sh_deg, src)
    return src


def struct_bytes(s):
    import ctypes
    return bytes(ctypes.string_at(ctypes.byref(s), ctypes.sizeof(s)))


def run_preprocess(src, bindings, n):
    """Binds the buffers, dispatches ceil(n / 256) workgroups (renderer.rs:411-413), returns the written buffers."""
    m = W.Module(src)
    out = {}
    for name, data in bindings.items():
        out[name] = m.bind(name, data)
    splat_size = m.struct_size("Splat")
    out["points_2d"] = m.bind("points_2d", bytearray(splat_size * n))
    out["sort_depths"] = m.bind("sort_depths", bytearray(4 * n))
    out["sort_indices"] = m.bind("sort_indices", bytearray(4 * n))
    out["sort_infos"] = m.bind("sort_infos", bytearray(m.struct_size("SortInfos")))
    out["sort_dispatch"] = m.bind("sort_dispatch", bytearray(m.struct_size("DispatchIndirect")))
    src_index = []

    def watch(gid):  # which invocation drew which store index (keys_size is SortInfos' first word)
        if int.from_bytes(out["sort_infos"][0:4], "little") > len(src_index):
            src_index.append(gid)
    m.dispatch("preprocess", (n + 255) // 256, watch)
    v = int(np.frombuffer(out["sort_infos"], dtype=np.uint32)[0])
    assert v == len(src_index)
    idx = np.frombuffer(out["sort_indices"], dtype=np.uint32)[:v]
    assert np.array_equal(idx, np.arange(v, dtype=np.uint32))
    return dict(num_visible=np.uint32(v), src_index=np.array(src_index, dtype=np.uint32),
                splats=np.frombuffer(out["points_2d"], dtype=np.uint8)[:v * splat_size].reshape(v, splat_size).copy(),
                keys=np.frombuffer(out["sort_depths"], dtype=np.uint32)[:v].copy(),
                dispatch_x=np.frombuffer(out["sort_dispatch"], dtype=np.uint32)[0])


# ---- K1 -------------------------------------------------------------------------------------------------------------
def k1_case(name):
    import oracle_lib as oracle
    import websplat as ws
    import wgsl_cases
    sc = wgsl_cases.k1_scene(ws, oracle, name)
    n, sh_deg = sc.gpc.num_points, sc.sh_deg
    # the uniforms as the ORACLE's host math builds them (camera.rs / renderer.rs restated in ws_oracle.c)
    cam = sc.args.camera
    ocam = oracle.make_camera(cam.position, cam.rotation, cam.fovx, cam.fovy, cam.znear, cam.zfar, cam.fov2view_ratio)
    cu = oracle.camera_uniform(ocam, *sc.viewport)
    a = sc.args
    rs = oracle.settings_uniform(oracle.make_aabb(sc.gpc.aabb.min, sc.gpc.aabb.max), sc.gpc.center,
                                 gaussian_scaling=a.gaussian_scaling, max_sh_deg=a.max_sh_deg,
                                 mip_splatting=a.mip_splatting, kernel_size=a.kernel_size,
                                 clipping_box=a.clipping_box, walltime=a.walltime, scene_extend=a.scene_extend,
                                 pc_mip=sc.gpc.mip_splatting, pc_kernel_size=sc.gpc.kernel_size)
    g = np.ascontiguousarray(sc.gpc.gaussians).view(np.uint8).reshape(n, -1)
    sh = np.ascontiguousarray(sc.gpc.sh_coefs).view(np.uint8).reshape(n, -1)
    cu_b, rs_b = struct_bytes(cu), struct_bytes(rs)
    res = run_preprocess(shader("preprocess.wgsl", sh_deg),
                         dict(camera=cu_b, render_settings=rs_b, gaussians=g.tobytes(), sh_coefs=sh.tobytes()), n)
    res.update(gaussians=g, sh_coefs=sh, camera_uniform=np.frombuffer(cu_b, dtype=np.uint8),
               settings_uniform=np.frombuffer(rs_b, dtype=np.uint8), sh_deg=np.uint32(sh_deg),
               viewport=np.array(sc.viewport, dtype=np.uint32))
    return res


# ---- K1c ------------------------------------------------------------------------------------------------------------
def k1c_case(name):
    import oracle_lib as oracle
    import websplat as ws
    import wgsl_cases
    gpc, cam, viewport, sh_deg = wgsl_cases.k1c_inputs(ws, name)
    n = gpc.num_points
    ocam = oracle.make_camera(cam.position, cam.rotation, cam.fovx, cam.fovy, cam.znear, cam.zfar, cam.fov2view_ratio)
    cu = oracle.camera_uniform(ocam, *viewport)
    rs = oracle.settings_uniform(oracle.make_aabb(gpc.aabb.min, gpc.aabb.max), gpc.center, max_sh_deg=sh_deg,
                                 pc_mip=gpc.mip_splatting, pc_kernel_size=gpc.kernel_size)
    q = gpc.quantization
    oq = oracle.make_quantization({k: (getattr(q, k).zero_point, getattr(q, k).scale)
                                   for k in ("color_dc", "color_rest", "opacity", "scaling_factor")})
    g = np.ascontiguousarray(gpc.gaussians).view(np.uint8).reshape(n, -1)
    shb = np.ascontiguousarray(gpc.sh_coefs).view(np.uint8).reshape(-1)
    cov = np.ascontiguousarray(gpc.covars).view(np.uint8).reshape(-1)
    pad = (-len(shb)) % 4 + 4  # the shader reads whole u32 words, one past the last coefficient's word
    cu_b, rs_b, q_b = struct_bytes(cu), struct_bytes(rs), struct_bytes(oq)
    res = run_preprocess(shader("preprocess_compressed.wgsl", sh_deg),
                         dict(camera=cu_b, render_settings=rs_b, vertices=g.tobytes(),
                              sh_coefs=shb.tobytes() + bytes(pad), geometries=cov.tobytes(), quantization=q_b), n)
    res.update(gaussians=g, sh_coefs=shb, covars=cov, camera_uniform=np.frombuffer(cu_b, dtype=np.uint8),
               settings_uniform=np.frombuffer(rs_b, dtype=np.uint8), quantization=np.frombuffer(q_b, dtype=np.uint8),
               sh_deg=np.uint32(sh_deg), viewport=np.array(viewport, dtype=np.uint32))
    return res


# ---- K6: vertex + fragment functions ----------------------------------------------------------------------------------
def k6_fragments(source="k1_default"):
    """For S splats of a K1 fixture and the pixel centres of a window around each: vs_main's four vertices, the
    interpolated screen_pos at the pixel, fs_main's premultiplied output (or discard).
    source "k1_default": 48 random splats.  source "frame_opaque" (-> wgsl_k6_fragments_opaque.npz): every splat of that
    fixture whose alpha is exactly 1.0 (wgsl_cases.opaque_rows) plus 10 random ones, and besides the thinned window EVERY
    pixel centre within two pixels of the splat's centre -- the fragments that reach `min(0.99, .)` (gaussian.wgsl:65)."""
    k1 = np.load(os.path.join(HERE, "wgsl_%s.npz" % source))
    w, h = (int(x) for x in k1["viewport"])
    splats = k1["splats"]
    rng = np.random.default_rng(5)
    dense = source != "k1_default"
    if dense:
        alpha_one = np.nonzero(splats.view(np.uint16).reshape(len(splats), -1)[:, 9] == 0x3C00)[0]
        others = np.setdiff1d(np.arange(len(splats)), alpha_one)
        pick = np.sort(np.concatenate([alpha_one, rng.choice(others, size=10, replace=False)]))
    else:
        pick = np.sort(rng.choice(len(splats), size=48, replace=False))
    m = W.Module(shader("gaussian.wgsl"))
    m.bind("points_2d", splats.tobytes())
    m.bind("indices", np.arange(len(splats), dtype=np.uint32).tobytes())
    rec_splat, rec_px, rec_pos, rec_out, rec_keep, verts = [], [], [], [], [], []
    for s in pick:
        vo = [m.invoke("vs_main", [W.u32(k), W.u32(int(s))]) for k in range(4)]
        P = np.array([[float(c) for c in v.f["position"].c[:2]] for v in vo], dtype=np.float64)   # NDC of the 4 vertices
        Q = np.array([[float(c) for c in v.f["screen_pos"].c] for v in vo], dtype=np.float64)     # their screen_pos
        verts.append(np.concatenate([P, Q], axis=1))
        # affine map NDC -> screen_pos through vertices 0 (+,+), 1 (-,+), 2 (+,-): a triangle strip of two triangles
        # that share it (the quad is a parallelogram), i.e. what the rasteriser's interpolation evaluates
        A = np.stack([P[1] - P[0], P[2] - P[0]], axis=1)
        if abs(np.linalg.det(A)) < 1e-30:
            continue
        Ainv = np.linalg.inv(A)
        dQ = np.stack([Q[1] - Q[0], Q[2] - Q[0]], axis=1)
        # pixel window: the quad's bounding box in pixels, thinned to at most ~60 pixels, plus everything within one
        # pixel of the cut-off circle is kept by the thinning (the interesting ones)
        px = (P[:, 0] * 0.5 + 0.5) * w
        py = (0.5 - P[:, 1] * 0.5) * h
        x0, x1 = int(np.floor(px.min())) - 1, int(np.ceil(px.max())) + 1
        y0, y1 = int(np.floor(py.min())) - 1, int(np.ceil(py.max())) + 1
        xs = np.arange(max(x0, 0), min(x1, w - 1) + 1)
        ys = np.arange(max(y0, 0), min(y1, h - 1) + 1)
        if len(xs) == 0 or len(ys) == 0:
            continue
        stride = max(1, int(np.sqrt(len(xs) * len(ys) / 60.0)))
        color = vo[0].f["color"]
        pixels = [(x, y) for y in ys[::stride] for x in xs[::stride]]
        if dense:  # every pixel centre within two pixels of the splat's centre (the mean of the quad's corners)
            cx, cy = px.mean(), py.mean()
            near = [(x, y) for y in range(int(cy) - 2, int(cy) + 3) for x in range(int(cx) - 2, int(cx) + 3)
                    if 0 <= x < w and 0 <= y < h]
            pixels = sorted(set(pixels) | set(near), key=lambda q: (q[1], q[0]))
        for (x, y) in pixels:
            ndc = np.array([(x + 0.5) / w * 2.0 - 1.0, 1.0 - (y + 0.5) / h * 2.0])
            st = Ainv @ (ndc - P[0])
            sp = Q[0] + dQ @ st
            spv = W.Vec([W.F32(sp[0]), W.F32(sp[1])])
            frag = W.StructVal("VertexOutput", dict(position=W.Vec([W.F32(x + 0.5), W.F32(y + 0.5), W.F32(0), W.F32(1)]),
                                                    screen_pos=spv, color=color))
            try:
                o = m.invoke("fs_main", [frag])
                rec_out.append([float(c) for c in o.c])
                rec_keep.append(1)
            except W.Discard:
                rec_out.append([0.0] * 4)
                rec_keep.append(0)
            rec_splat.append(int(s))
            rec_px.append((x, y))
            rec_pos.append((float(spv.c[0]), float(spv.c[1])))
    return dict(viewport=np.array([w, h], dtype=np.uint32), splats=splats, picked=pick.astype(np.uint32),
                vertices=np.array(verts, dtype=np.float32), frag_splat=np.array(rec_splat, dtype=np.uint32),
                frag_pixel=np.array(rec_px, dtype=np.uint32), frag_screen_pos=np.array(rec_pos, dtype=np.float32),
                frag_out=np.array(rec_out, dtype=np.float32), frag_keep=np.array(rec_keep, dtype=np.uint8))


# ---- a whole frame: K1 -> stable sort by key -> instanced draw with premultiplied "over" ----------------------------------
def frame(name="frame"):
    """preprocess.wgsl, then the draw of renderer.rs:240-283: instances in ascending key order (stable: equal keys keep
    their store order, gpu_rs.rs), vs_main / fs_main from source for every pixel centre inside an instance's quad, and
    the pipeline's blend state PREMULTIPLIED_ALPHA_BLENDING (renderer.rs:65: dst = src + dst * (1 - src.a), f32 here)
    on a target cleared to the case's background (transparent for `frame`, opaque for `frame_opaque`).  The kept disc (radius sqrt(2 CUTOFF) in screen_pos units) lies strictly inside the quad
    (half-width CUTOFF), so the rasteriser's edge rules never decide a pixel."""
    import wgsl_cases
    res = k1_case(name)
    w, h = (int(x) for x in res["viewport"])
    splats, keys = res["splats"], res["keys"]
    order = np.argsort(keys, kind="stable").astype(np.uint32)
    m = W.Module(shader("gaussian.wgsl"))
    m.bind("points_2d", splats.tobytes())
    m.bind("indices", order.tobytes())
    background = np.array(wgsl_cases.FRAME_BACKGROUND[name], dtype=np.float32)
    img = np.empty((h, w, 4), dtype=np.float32)
    img[:] = background   # begin_render_pass: LoadOp::Clear(background)
    one = np.float32(1.0)
    nfrag = 0
    for inst in range(len(order)):
        vo = [m.invoke("vs_main", [W.u32(k), W.u32(inst)]) for k in range(4)]
        P = np.array([[float(c) for c in v.f["position"].c[:2]] for v in vo], dtype=np.float64)
        Q = np.array([[float(c) for c in v.f["screen_pos"].c] for v in vo], dtype=np.float64)
        A = np.stack([P[1] - P[0], P[2] - P[0]], axis=1)
        if abs(np.linalg.det(A)) < 1e-30:
            continue
        Ainv = np.linalg.inv(A)
        dQ = np.stack([Q[1] - Q[0], Q[2] - Q[0]], axis=1)
        px = (P[:, 0] * 0.5 + 0.5) * w
        py = (0.5 - P[:, 1] * 0.5) * h
        xs = range(max(int(np.floor(px.min())), 0), min(int(np.ceil(px.max())), w - 1) + 1)
        ys = range(max(int(np.floor(py.min())), 0), min(int(np.ceil(py.max())), h - 1) + 1)
        color = vo[0].f["color"]
        for y in ys:
            for x in xs:
                ndc = np.array([(x + 0.5) / w * 2.0 - 1.0, 1.0 - (y + 0.5) / h * 2.0])
                st = Ainv @ (ndc - P[0])
                if st[0] < 0 or st[0] > 1 or st[1] < 0 or st[1] > 1:
                    continue  # outside the quad
                sp = Q[0] + dQ @ st
                frag = W.StructVal("VertexOutput", dict(position=W.Vec([W.F32(x + 0.5), W.F32(y + 0.5), W.F32(0), W.F32(1)]),
                                                        screen_pos=W.Vec([W.F32(sp[0]), W.F32(sp[1])]), color=color))
                try:
                    o = m.invoke("fs_main", [frag])
                except W.Discard:
                    continue
                src = np.array([c for c in o.c], dtype=np.float32)
                img[y, x] = src + img[y, x] * (one - src[3])
                nfrag += 1
    res.update(order=order, image=img, fragments=np.uint32(nfrag), background=background)
    return res


# ---- the radix sort (gpu_rs.rs + radix_sort.wgsl) ------------------------------------------------------------------------
def sort_case(n, seed, subgroup=1):
    """GPURSSorter::record_sort (gpu_rs.rs:865-873) on n seeded (key, payload) pairs, kernels from radix_sort.wgsl.

    The shader is instantiated as new_with_sg_size does (gpu_rs.rs:177-256: eleven constants in front, three placeholders
    replaced) with subgroup size 1 -- one of the sizes the reference's own search tries (gpu_rs.rs:65-139).  It is the
    race-free instantiation: for larger sizes `scatter` relies on the lanes of a subgroup running in lock-step between an
    atomicStore and the neighbours' atomicLoad (no barrier), which this interpreter -- threads that meet at barriers --
    does not model.  The sorted order does not depend on that size.  Buffers, GeneralInfo and dispatch sizes follow
    create_keyval_buffers / create_internal_mem_buffer / create_bind_group / get_scatter_histogram_sizes
    (gpu_rs.rs:478-663) and record_calculate_histogram / record_prefix_histogram / record_scatter_keys (:729-835); the
    payload buffers have keysize elements only (the reference's under-allocation), so the padded tail is read and
    written out of bounds, which WebGPU's robust buffer access turns into zeros / dropped stores."""
    WG, LOG2, RADIX, KEYVAL, ROWS, PREFIX_WG, SCATTER_WG = 256, 8, 256, 4, 15, 128, 256
    sweep0 = RADIX // subgroup
    sweep1 = sweep0 // subgroup
    consts = [subgroup, WG, LOG2, RADIX, KEYVAL, ROWS, ROWS, RADIX + ROWS * SCATTER_WG, 0, sweep0, sweep0 + sweep1]
    names = ["histogram_sg_size", "histogram_wg_size", "rs_radix_log2", "rs_radix_size", "rs_keyval_size",
             "rs_histogram_block_rows", "rs_scatter_block_rows", "rs_mem_dwords", "rs_mem_sweep_0_offset",
             "rs_mem_sweep_1_offset", "rs_mem_sweep_2_offset"]
    src = "".join("const %s: u32 = %du;\n" % (k, v) for k, v in zip(names, consts)) + shader("radix_sort.wgsl")
    src = src.replace("{histogram_wg_size}", str(WG)).replace("{prefix_wg_size}", str(PREFIX_WG)).replace("{scatter_wg_size}", str(SCATTER_WG))
    block = WG * ROWS
    scatter_blocks = (n + block - 1) // block
    count_ru_scatter = scatter_blocks * block
    histo_blocks = (count_ru_scatter + block - 1) // block
    padded = histo_blocks * block                                # GeneralInfo::padded_size
    keybuf_elems = ((n + block) // block + 1) * block            # create_keyval_buffers
    rng = np.random.default_rng(seed)
    # depth-like float keys with many ties, a few extreme bit patterns, payload = identity (preprocess.wgsl:275)
    z = np.round(rng.uniform(0.5, 40.0, size=n).astype(np.float32) * np.float32(8.0)) / np.float32(8.0)
    keys = z.astype(np.float32).view(np.uint32).copy()
    keys[:6] = [0, 1, 0x7F7FFFFF, 0x80000000, 0xFFFFFFFE, 0x00800000]
    payload = np.arange(n, dtype=np.uint32)
    m = W.Module(src)
    m.robust = True
    ka = np.zeros(keybuf_elems, dtype=np.uint32)
    ka[:n] = keys
    bufs = dict(infos=m.bind("infos", np.array([n, padded, 4, 0, 0], dtype=np.uint32).tobytes()),
                histograms=m.bind("histograms", bytes((KEYVAL + scatter_blocks - 1 + 1) * RADIX * 4)),
                keys=m.bind("keys", ka.tobytes()), keys_b=m.bind("keys_b", bytes(keybuf_elems * 4)),
                payload_a=m.bind("payload_a", payload.tobytes()), payload_b=m.bind("payload_b", bytes(max(n * 4, 1))))
    m.dispatch_workgroups("zero_histograms", histo_blocks, threads=False)
    m.dispatch_workgroups("calculate_histogram", histo_blocks)
    m.dispatch_workgroups("prefix_histogram", 4)
    for entry in ("scatter_even", "scatter_odd", "scatter_even", "scatter_odd"):
        m.dispatch_workgroups(entry, scatter_blocks)
    out_k = np.frombuffer(bufs["keys"], dtype=np.uint32)[:n].copy()
    out_p = np.frombuffer(bufs["payload_a"], dtype=np.uint32)[:n].copy()
    return dict(keys_in=keys, payload_in=payload, keys_out=out_k, payload_out=out_p, subgroup=np.uint32(subgroup),
                padded_size=np.uint32(padded), scatter_blocks=np.uint32(scatter_blocks))


import wgsl_cases  # noqa: E402

CASES = {}
for c in wgsl_cases.K1_CASES:
    CASES["k1_" + c] = (lambda c=c: k1_case(c))
for c in wgsl_cases.K1C_CASES:
    CASES["k1c_" + c] = (lambda c=c: k1c_case(c))
CASES["k6_fragments"] = k6_fragments
for c in wgsl_cases.FRAME_CASES:
    CASES[c] = (lambda c=c: frame(c))
CASES["k6_fragments_opaque"] = lambda: k6_fragments("frame_opaque")   # (reads wgsl_frame_opaque.npz: generated after it)
CASES["sort_small"] = lambda: sort_case(700, 11)
CASES["sort_two_blocks"] = lambda: sort_case(5000, 12)


def main():
    want = sys.argv[1:] or list(CASES)
    for name in want:
        t0 = time.time()
        res = CASES[name]()
        path = os.path.join(HERE, "wgsl_%s.npz" % name)
        np.savez_compressed(path, **res)
        extra = ("V = %d" % int(res["num_visible"]) if "num_visible" in res else "") + \
                ("%d pairs" % len(res["keys_in"]) if "keys_in" in res else "") + \
                (" %d fragments" % (int(res["fragments"]) if "fragments" in res else len(res["frag_keep"]))
                 if ("fragments" in res or "frag_keep" in res) else "")
        print("%-16s %s  %.1f s  %d bytes" % (name, extra, time.time() - t0, os.path.getsize(path)))


if __name__ == "__main__":
    main()
