"""Does the device-side binning decision pick the faster granularity?  N Gaussians x resolution (bonsai-like synthetic
distribution), 4 frames in flight: frames/s with the binning tile forced to the compositing tile (WS_BIN_SHIFT=0), forced to
2 x 2 of them (=1) and decided per frame on the device (auto), plus what auto chose.  Writes gpurun_out/sweep_bin.json."""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "web-splat_amd"), os.path.join(ROOT, "tests"), ROOT]
import numpy as np, torch
import websplat as ws
from websplat import synth

STREAMS = [torch.cuda.Stream() for _ in range(4)]
STEPS = int(os.environ.get("STEPS", "300"))
out = []
for n in (250_000, 500_000, 1_000_000, 2_000_000, 5_000_000):
    gpc = ws.GenericGaussianPointCloud.from_ply_rows(synth.scene_c2(n=n, seed=1), 3)
    for (w, h) in ((800, 600), (1200, 799), (1920, 1080), (3840, 2160)):
        f = 1200.0 * w / 1200.0
        row = {"gaussians": n, "width": w, "height": h}
        for mode in ("0", "1", "auto"):
            os.environ["WS_BIN_SHIFT"] = mode  # read when the context is created
            ctx = ws.Context(0)
            pc = ws.PointCloud(ctx, gpc)
            views = []
            for cj in synth.orbit_cameras(16, w, h, f, f):
                cam = ws.PerspectiveCamera.from_scene_camera(cj.position, cj.rotation, cj.fx, cj.fy, w, h)
                cam.fit_near_far(gpc.aabb)
                views.append(ws.SplattingArgs(camera=cam, viewport=(w, h), max_sh_deg=3))
            rs = [ws.GaussianRenderer(ctx, "rgba32float", 3, False) for _ in range(4)]
            tg = [torch.empty((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(4)]
            st = [s_.cuda_stream for s_ in STREAMS]
            def frame(i):
                k = i % 4
                rs[k].prepare(pc, views[i % 16], stream=st[k]); rs[k].render(pc, target_ptr=tg[k].data_ptr(), stream=st[k])
            for i in range(32): frame(i)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(STEPS): frame(i)
            torch.cuda.synchronize()
            row[f"fps_{mode}"] = STEPS / (time.perf_counter() - t0)
            fs = rs[0].frame_stats()
            row[f"entries_{mode}"] = fs["num_tile_entries"]
            if mode == "auto":
                row["auto_tile"] = rs[0].binning_tile()[0]
                row["visible"] = fs["num_visible"]
            assert fs["overflow"] == 0
            for r in rs: r.close()
            del tg
            pc.close()
            ctx.close()
        best = "1" if row["fps_1"] > row["fps_0"] else "0"
        row["ratio_entries_0_over_1"] = row["entries_0"] / max(row["entries_1"], 1)
        row["auto_is_best"] = (row["auto_tile"] == 64) == (best == "1")
        row["auto_vs_best"] = row["fps_auto"] / max(row["fps_0"], row["fps_1"])
        out.append(row)
        print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in row.items()}, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "sweep_bin.json"), "w"), indent=1)
print("auto picked the faster granularity in", sum(r["auto_is_best"] for r in out), "of", len(out), "cells; worst auto/best =",
      round(min(r["auto_vs_best"] for r in out), 3))
