"""Frames/s over N Gaussians x resolution (bonsai-like synthetic distribution), 4 frames in flight and 1 in flight.
Writes gpurun_out/sweep.json; copy to profiles/ to keep."""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # four frames in flight need four hardware queues (bench.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "web-splat_amd"), os.path.join(ROOT, "tests"), ROOT]
import numpy as np, torch
import websplat as ws
from websplat import synth

ctx = ws.Context(0)
out = []
# four real streams, created ONCE: a renderer on the legacy NULL stream serialises against the others (the round-1 sweeps
# had one), and streams taken from torch's pool again and again end up sharing hardware queues in some cells
STREAMS = [torch.cuda.Stream() for _ in range(4)]
for n in (250_000, 500_000, 1_000_000, 2_000_000, 5_000_000):
    gpc = ws.GenericGaussianPointCloud.from_ply_rows(synth.scene_c2(n=n, seed=1), 3)
    pc = ws.PointCloud(ctx, gpc)
    for (w, h) in ((800, 600), (1200, 799), (1920, 1080), (3840, 2160)):
        f = 1200.0 * w / 1200.0
        views = []
        for cj in synth.orbit_cameras(16, w, h, f, f):
            cam = ws.PerspectiveCamera.from_scene_camera(cj.position, cj.rotation, cj.fx, cj.fy, w, h)
            cam.fit_near_far(gpc.aabb)
            views.append(ws.SplattingArgs(camera=cam, viewport=(w, h), max_sh_deg=3))
        row = {"gaussians": n, "width": w, "height": h}
        for ns in (4, 1):
            rs = [ws.GaussianRenderer(ctx, "rgba32float", 3, False) for _ in range(ns)]
            tg = [torch.empty((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(ns)]
            st = [s_.cuda_stream for s_ in STREAMS[:ns]]
            def frame(i):
                k = i % ns
                rs[k].prepare(pc, views[i % 16], stream=st[k]); rs[k].render(pc, target_ptr=tg[k].data_ptr(), stream=st[k])
            for i in range(48): frame(i)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            steps = 600
            for i in range(steps): frame(i)
            torch.cuda.synchronize()
            row[f"fps_{ns}_in_flight"] = steps / (time.perf_counter() - t0)
            if ns == 1:
                fs = rs[0].frame_stats()
                row["visible"], row["tile_entries"], row["overflow"] = fs["num_visible"], fs["num_tile_entries"], fs["overflow"]
            for r in rs: r.close()
            del tg
        out.append(row)
        print(row, flush=True)
    pc.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "sweep.json"), "w"), indent=1)
