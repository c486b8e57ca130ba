#!/usr/bin/env python3
"""Render views of a bench workload with the library WEBSPLAT_LIB selects and write the raw images + frame counters:
the A/B equality check between library builds (the image must not change when only the binning / staging changes).
  python scripts/dump_images.py <workload> <outdir> [views...]     then     python scripts/dump_images.py --compare A B
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "web-splat_amd"), os.path.join(ROOT, "tests"), ROOT]
import numpy as np  # noqa: E402


def compare(a, b):
    ok = True
    for f in sorted(os.listdir(a)):
        if not f.endswith(".npy"):
            continue
        x, y = np.load(os.path.join(a, f)), np.load(os.path.join(b, f))
        same = x.shape == y.shape and np.array_equal(x.view(np.uint32), y.view(np.uint32))
        d = float(np.abs(x.astype(np.float64) - y.astype(np.float64)).max()) if x.shape == y.shape else float("nan")
        print(f"{f}: {'IDENTICAL' if same else 'DIFFERENT'} max-abs {d:.3e} differing pixels "
              f"{int((x != y).any(axis=-1).sum()) if x.shape == y.shape else -1}")
        ok &= same
    for f in sorted(os.listdir(a)):
        if f.endswith(".json"):
            print(f, json.load(open(os.path.join(a, f))), "|", json.load(open(os.path.join(b, f))))
    return 0 if ok else 1


def main():
    if sys.argv[1] == "--compare":
        sys.exit(compare(sys.argv[2], sys.argv[3]))
    name, out = sys.argv[1], sys.argv[2]
    view_ids = [int(v) for v in sys.argv[3:]] or [0, 3]
    os.makedirs(out, exist_ok=True)
    import bench
    import websplat as ws
    ctx = ws.Context(0)
    gpc, views, (w, h), _ = bench.build_workload(ws, name, max(view_ids) + 1)
    pc = ws.PointCloud(ctx, gpc)
    r = ws.GaussianRenderer(ctx, os.environ.get("WS_DUMP_FORMAT", "rgba32float"), gpc.sh_deg, gpc.compressed)
    for vi in view_ids:
        r.prepare(pc, views[vi])
        r.render(pc, background=(0.1, 0.2, 0.3, 1.0) if vi % 2 else (0, 0, 0, 0))
        img = r.download_target()
        st = r.frame_stats()
        np.save(os.path.join(out, f"{name}_view{vi}.npy"), img)
        json.dump({k: int(st[k]) for k in ("num_visible", "num_tile_entries", "overflow")},
                  open(os.path.join(out, f"{name}_view{vi}.json"), "w"))
    r.close()
    pc.close()
    ctx.close()


if __name__ == "__main__":
    main()
