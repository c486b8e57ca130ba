"""Second probe of the slow mode (scripts/slowmode_probe.py found: it follows the upload of a point cloud).  Which part of an
upload triggers it -- the allocation, the free, the copy from pageable memory, the copy from pinned memory -- and how long does
it last?  After each trigger: frames/s of consecutive windows of 250 frames."""
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
rs = [ws.GaussianRenderer(ctx, "rgba32float", 3, False) for _ in range(4)]
A = [torch.cuda.Stream() for _ in range(4)]
st = [s.cuda_stream for s in A]
tg = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(4)]
def frame(i):
    k = i % 4
    rs[k].prepare(pc, views[i % 16], stream=st[k]); rs[k].render(pc, target_ptr=tg[k].data_ptr(), stream=st[k])
def windows(tag, n=6, frames=250):
    out = []
    t_begin = time.perf_counter()
    for w in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(frames): frame(i)
        torch.cuda.synchronize()
        out.append(frames / (time.perf_counter() - t0))
    print(f"{tag:64s} " + " ".join(f"{x:6.0f}" for x in out) + f"   ({time.perf_counter() - t_begin:.2f} s)", flush=True)
for i in range(64): frame(i)
windows("baseline")
MB = 64
src_pageable = np.random.default_rng(0).integers(0, 255, MB << 20, dtype=np.uint8)
t_page = torch.from_numpy(src_pageable)
t_pin = torch.from_numpy(src_pageable.copy()).pin_memory()
d = torch.empty(MB << 20, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
windows("after torch.empty (device allocation through torch's cache)")
d.copy_(t_pin); torch.cuda.synchronize()
windows("after a 64 MB copy from PINNED host memory")
d.copy_(t_page); torch.cuda.synchronize()
windows("after a 64 MB copy from PAGEABLE host memory")
p = ctx.malloc(MB << 20)
windows("after hipMalloc of 64 MB (library)")
ctx.free(p)
windows("after hipFree of it")
pc2 = ws.PointCloud(ctx, gpc)
windows("after a second PointCloud (hipMalloc + hipMemcpy from pageable)")
pc2.close()
windows("after closing it (hipFree)")
d.copy_(t_page); torch.cuda.synchronize()
windows("after another pageable copy", n=12)
