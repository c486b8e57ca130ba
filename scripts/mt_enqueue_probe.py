#!/usr/bin/env python3
"""Does submitting frames from several host threads lift the host-bound cells?  (small scenes: the GPU needs less time per
frame than ONE thread needs to enqueue its 22 launches.)  F one-slot view batches, each driven by its own Python thread
(ctypes releases the GIL in the library call), against one F-slot batch driven by one thread.
    python scripts/mt_enqueue_probe.py [n_gaussians] [w] [h]
"""
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "web-splat_amd"), ROOT]
import torch  # noqa: E402
import websplat as ws  # noqa: E402
from websplat import synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 800
    h = int(sys.argv[3]) if len(sys.argv) > 3 else 600
    F, frames = 4, 2000
    ctx = ws.Context(0)
    gpc = ws.GenericGaussianPointCloud.from_ply_rows(synth.scene_c2(n=n, seed=1) if n > 20_000 else synth.scene_c1(n=n, seed=0), 3)
    pc = ws.PointCloud(ctx, gpc)
    views = []
    for cj in synth.orbit_cameras(16, w, h, float(w), float(w)):
        cam = ws.PerspectiveCamera.from_scene_camera(cj.position, cj.rotation, cj.fx, cj.fy, w, h)
        cam.fit_near_far(gpc.aabb)
        views.append(ws.SplattingArgs(camera=cam, viewport=(w, h), max_sh_deg=3))
    packed = ws.ViewBatch.pack_views(views)
    import ctypes as C
    targets = [torch.empty((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(F)]

    def plan(count, slot_targets):
        arr = (type(packed[0]) * count)()
        ptrs = (C.c_void_p * count)()
        for j in range(count):
            arr[j] = packed[j % len(views)]
            ptrs[j] = slot_targets[j % len(slot_targets)].data_ptr()
        return arr, ptrs
    # one thread, one F-slot batch
    one = ws.ViewBatch(ctx, "rgba32float", 3, False, F)
    p = plan(frames, targets)
    one.render(pc, *plan(64, targets), w * 16)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    one.render(pc, p[0], p[1], w * 16)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"N {n} {w}x{h}: ONE thread, {F} slots: {frames / el:9.1f} frames/s, enqueue {t_enq / frames * 1e6:6.1f} us/frame, errors {one.errors()}")
    one.close()
    # F threads, F one-slot batches
    batches = [ws.ViewBatch(ctx, "rgba32float", 3, False, 1) for _ in range(F)]
    plans = [plan(frames // F, [targets[k]]) for k in range(F)]
    for k in range(F):
        batches[k].render(pc, *plan(16, [targets[k]]), w * 16)
    torch.cuda.synchronize()
    t_thread = [0.0] * F

    def work(k):
        ta = time.perf_counter()
        batches[k].render(pc, plans[k][0], plans[k][1], w * 16)
        t_thread[k] = time.perf_counter() - ta
    th = [threading.Thread(target=work, args=(k,)) for k in range(F)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    errs = [b.errors() for b in batches]
    print(f"N {n} {w}x{h}: {F} threads, {F} x 1 slot: {frames / el:9.1f} frames/s, enqueue wall {t_enq / frames * 1e6:6.1f} us/frame "
          f"(per thread {max(t_thread) / (frames // F) * 1e6:6.1f} us/frame), errors {errs}")
    for b in batches:
        b.close()
    pc.close()
    ctx.close()


if __name__ == "__main__":
    main()
