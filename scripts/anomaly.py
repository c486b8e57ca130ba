import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "web-splat_amd"), os.path.join(ROOT, "tests"), ROOT]
import numpy as np, torch
import websplat as ws
from websplat import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (800, 600)
ctx = ws.Context(0)
gpc = ws.GenericGaussianPointCloud.from_ply_rows(synth.scene_c2(n=n, seed=1), 3)
pc = ws.PointCloud(ctx, gpc)
f = 1200.0 * w / 1200.0
views = []
for cj in synth.orbit_cameras(16, w, h, f, f):
    cam = ws.PerspectiveCamera.from_scene_camera(cj.position, cj.rotation, cj.fx, cj.fy, w, h)
    cam.fit_near_far(gpc.aabb)
    views.append(ws.SplattingArgs(camera=cam, viewport=(w, h), max_sh_deg=3))
for ns in (1, 2, 4):
    rs = [ws.GaussianRenderer(ctx, "rgba32float", 3, False) for _ in range(ns)]
    tg = [torch.empty((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(ns)]
    st = [torch.cuda.current_stream().cuda_stream] + [torch.cuda.Stream().cuda_stream for _ in range(ns - 1)]
    def frame(i):
        k = i % ns
        rs[k].prepare(pc, views[i % 16], stream=st[k]); rs[k].render(pc, target_ptr=tg[k].data_ptr(), stream=st[k])
    for i in range(16): frame(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(160): frame(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"cut={os.environ.get('WS_DEBUG_CUT','0')} streams {ns}: {1e6*dt/160:.1f} us/frame  stats", [r.frame_stats() for r in rs][:2], flush=True)
    for r in rs: r.close()
