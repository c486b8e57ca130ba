"""What puts a long-lived process into the slow mode of scripts/sweep*.py (four frames in flight running at the one-frame rate)?
Phases on ONE scene (250 k Gaussians, 800x600), 1000 frames each; prints frames/s per phase."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "web-splat_amd"), os.path.join(ROOT, "tests"), ROOT]
import numpy as np, torch
import websplat as ws
from websplat import synth

N, W, H = int(os.environ.get("N", "250000")), 800, 600
gpc = ws.GenericGaussianPointCloud.from_ply_rows(synth.scene_c2(n=N, seed=1), 3)
ctx = ws.Context(0)
pc = ws.PointCloud(ctx, gpc)
views = []
for cj in synth.orbit_cameras(16, W, H, 800.0, 800.0):
    cam = ws.PerspectiveCamera.from_scene_camera(cj.position, cj.rotation, cj.fx, cj.fy, W, H)
    cam.fit_near_far(gpc.aabb)
    views.append(ws.SplattingArgs(camera=cam, viewport=(W, H), max_sh_deg=3))

def renderers(vp=(W, H)):
    return [ws.GaussianRenderer(ctx, "rgba32float", 3, False) for _ in range(4)]

def run(tag, rs, streams, tg, frames=1000):
    st = [s.cuda_stream for s in streams]
    def frame(i):
        k = i % 4
        rs[k].prepare(pc, views[i % 16], stream=st[k]); rs[k].render(pc, target_ptr=tg[k].data_ptr(), stream=st[k])
    for i in range(32): frame(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(frames): frame(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{tag:58s} {frames / (t2 - t0):8.0f} frames/s   host enqueue {1e6 * (t1 - t0) / frames:6.1f} us/frame", flush=True)

A = [torch.cuda.Stream() for _ in range(4)]
tg = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(4)]
rs = renderers()
run("P0 fresh process", rs, A, tg)
run("P1 again, nothing changed", rs, A, tg)
for r in rs: r.close()
rs = renderers()
run("P2 renderers closed and re-created, same streams", rs, A, tg)
B = [torch.cuda.Stream() for _ in range(4)]
run("P3 four NEW streams (old ones kept alive)", rs, B, tg)
run("P4 back on the first four streams", rs, A, tg)
for r in rs: r.close()
rs = renderers()
torch.cuda.synchronize(); time.sleep(1.0)
run("P5 renderers re-created, 1 s idle", rs, A, tg)
del tg
tg = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(4)]
run("P6 targets re-allocated", rs, A, tg)
pc2 = ws.PointCloud(ctx, gpc)
pc2.close()
run("P7 a second point cloud uploaded and freed", rs, A, tg)
ctx2 = ws.Context(0)
ctx2.close()
run("P8 a second context created and closed", rs, A, tg)
big = [ws.GaussianRenderer(ctx, "rgba32float", 3, False) for _ in range(2)]
cj = synth.orbit_cameras(1, 3840, 2160, 3840.0, 3840.0)[0]
cam = ws.PerspectiveCamera.from_scene_camera(cj.position, cj.rotation, cj.fx, cj.fy, 3840, 2160)
cam.fit_near_far(gpc.aabb)
a4k = ws.SplattingArgs(camera=cam, viewport=(3840, 2160), max_sh_deg=3)
for b in big:
    b.prepare(pc, a4k); b.render(pc)
ctx.sync()
for b in big: b.close()
run("P9 two 4K renderers used on the NULL stream and closed", rs, A, tg)
for r in rs: r.close()
rs = renderers()
run("P10 renderers re-created once more", rs, A, tg)
