"""One configuration of scripts/sweep.py with progress output: python scripts/sweep_one.py N W H [streams...]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "web-splat_amd"), os.path.join(ROOT, "tests"), ROOT]
import numpy as np, torch
import websplat as ws
from websplat import synth
n, w, h = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
streams = [int(x) for x in sys.argv[4:]] or [4, 1]
ctx = ws.Context(0)
if os.environ.get("PREHISTORY"):   # what precedes a cell inside scripts/sweep.py: bigger renderers created, used and destroyed
    for (pn, pw, ph) in ((250_000, 3840, 2160), (250_000, 1920, 1080)):
        g0 = ws.GenericGaussianPointCloud.from_ply_rows(synth.scene_c2(n=pn, seed=1), 3)
        p0 = ws.PointCloud(ctx, g0)
        cj = synth.orbit_cameras(1, pw, ph, float(pw), float(pw))[0]
        cam = ws.PerspectiveCamera.from_scene_camera(cj.position, cj.rotation, cj.fx, cj.fy, pw, ph)
        cam.fit_near_far(g0.aabb)
        a0 = ws.SplattingArgs(camera=cam, viewport=(pw, ph), max_sh_deg=3)
        r0 = [ws.GaussianRenderer(ctx, "rgba32float", 3, False) for _ in range(4)]
        for r in r0:
            r.prepare(p0, a0); r.render(p0)
        ctx.sync()
        for r in r0: r.close()
        p0.close()
gpc = ws.GenericGaussianPointCloud.from_ply_rows(synth.scene_c2(n=n, seed=1), 3)
pc = ws.PointCloud(ctx, gpc)
f = 1200.0 * w / 1200.0
views = []
for cj in synth.orbit_cameras(16, w, h, f, f):
    cam = ws.PerspectiveCamera.from_scene_camera(cj.position, cj.rotation, cj.fx, cj.fy, w, h)
    cam.fit_near_far(gpc.aabb)
    views.append(ws.SplattingArgs(camera=cam, viewport=(w, h), max_sh_deg=3))
for ns in streams:
    rs = [ws.GaussianRenderer(ctx, "rgba32float", 3, False) for _ in range(ns)]
    tg = [torch.empty((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(ns)]
    keep = [torch.cuda.Stream() for _ in range(ns)]
    st = [torch.cuda.current_stream().cuda_stream] + [s_.cuda_stream for s_ in keep[1:]]
    if os.environ.get("NONNULL"):
        st = [s_.cuda_stream for s_ in keep]
    print("streams", ns, "stream handles", st, flush=True)
    nframes = int(os.environ.get("FRAMES", "64"))
    torch.cuda.synchronize(); t_start = time.perf_counter()
    for i in range(nframes):
        k = i % ns
        rs[k].prepare(pc, views[i % 16], stream=st[k]); rs[k].render(pc, target_ptr=tg[k].data_ptr(), stream=st[k])
        mix = os.environ.get("MIX")
        if mix:
            torch.cuda.synchronize()
            if mix == "torchcpu": torch.zeros(4, device="cuda").cpu()
            if mix == "ctxdl":
                if i == 0: dbuf = ctx.malloc(64)
                ctx.download(dbuf, (4,), np.float32)
            if mix == "target": rs[k].download_target() if False else ctx.download(tg[k].data_ptr(), (4,), np.float32)
            print("frame", i, "mix ok", flush=True)
        if os.environ.get("SYNC_EACH"):
            torch.cuda.synchronize(); print("frame", i, "ok", rs[k].frame_stats() if os.environ.get("SYNC_EACH") == "1" else "", flush=True)
    torch.cuda.synchronize()
    print("streams %d: %.0f frames/s" % (ns, nframes / (time.perf_counter() - t_start)), flush=True)
    print("streams", ns, "done", rs[0].frame_stats(), rs[0].errors(), flush=True)
    for r in rs: r.close()
