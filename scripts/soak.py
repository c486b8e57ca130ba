"""Soak run (GPU): a long stream of frames in flight must stay bit-identical to its own first pass and must not drop data.

A view batch (4 frames in flight, `bench.py`'s path) draws the workload's 64 orbit views into 64 resident targets; the targets of
the first pass are digested; then the same 64 views are drawn ROUNDS more times back to back (the host's run-ahead window, the
slots' scratch, the demand / progress mailboxes and the per-frame zero arena all cycle thousands of times), half-way through a
single renderer of the same context draws a few lone frames in between (the workgroup-order policy flips and flips back), and the
targets of the last pass are digested again.  Pass = every digest equal, no sticky error bit, the device's visible / entry counts
of the last frame equal those of the first pass.

  python scripts/soak.py [workload=hd1m] [rounds=1500] [out.json]   -> gpurun_out/soak_<workload>.json
"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "web-splat_amd"), ROOT]
import numpy as np  # noqa: E402


def soak(workload="hd1m", rounds=1500, n_views=64, frames_in_flight=4, color_format="rgba32float"):
    import websplat as ws
    import bench
    ctx = ws.Context(0)
    gpc, views, (w, h), _ = bench.build_workload(ws, workload, n_views)
    pc = ws.PointCloud(ctx, gpc)
    vb = ws.ViewBatch(ctx, color_format, gpc.sh_deg, bool(getattr(gpc, "compressed", False)), frames_in_flight)
    texel = vb.texel_bytes
    pitch = w * texel
    targets = [ctx.malloc(h * pitch) for _ in range(n_views)]
    packed = ws.ViewBatch.pack_views(views)

    def digests():
        out = []
        for t in targets:
            a = ctx.download(t, (h, w * texel), np.uint8)
            out.append(hashlib.sha1(a.tobytes()).hexdigest())
        return out

    def counts():
        return [tuple(int(vb.renderer(s).frame_stats()[k]) for k in ("num_visible", "num_tile_entries"))
                for s in range(vb.frames_in_flight)]

    vb.render(pc, packed, targets, pitch)
    vb.sync()
    first = digests()
    first_counts = counts()
    err0 = vb.errors()
    lone = ws.GaussianRenderer(ctx, color_format, gpc.sh_deg, bool(getattr(gpc, "compressed", False)))
    t0 = time.perf_counter()
    for r in range(rounds):
        vb.render(pc, packed, targets, pitch)
        if r == rounds // 2:   # a lone renderer of the same context in between: the order policy sees a run on one stream
            vb.sync()
            for v in views[:6]:
                lone.prepare(pc, v)
                lone.render(pc)
            ctx.sync()
    vb.sync()
    dt = time.perf_counter() - t0
    last = digests()
    res = {"workload": workload, "frames": rounds * n_views, "frames_in_flight": vb.frames_in_flight, "seconds": dt,
           "frames_per_sec": rounds * n_views / dt if dt > 0 else None,
           "errors_first_pass": err0, "errors_after": vb.errors(), "lone_renderer_errors": lone.errors()[0],
           "digests_equal": first == last, "views_that_differ": [i for i, (a, b) in enumerate(zip(first, last)) if a != b],
           "counts_equal": counts() == first_counts,
           "distinct_images": len(set(first))}
    res["pass"] = bool(res["digests_equal"] and res["counts_equal"] and not res["errors_first_pass"] and not res["errors_after"]
                       and not res["lone_renderer_errors"])   # (distinct_images: c3's views share one camera -> 1)
    lone.close()
    for t in targets:
        ctx.free(t)
    vb.close()
    pc.close()
    ctx.close()
    return res


if __name__ == "__main__":
    wl = sys.argv[1] if len(sys.argv) > 1 else "hd1m"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", f"soak_{wl}.json")
    out = soak(wl, rounds)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out))
    sys.exit(0 if out["pass"] else 1)
