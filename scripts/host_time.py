"""Host-side cost of submitting one frame (22 launches + one memset) vs the GPU time of the frame.
Short bursts (16 frames into idle queues) so that the submission is not throttled by queue back-pressure."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path[:0] = ["web-splat_amd", "tests", "."]
import numpy as np, torch
import websplat as ws, bench
ctx = ws.Context(0)
gpc, views, (w, h), _ = bench.build_workload(ws, sys.argv[1] if len(sys.argv) > 1 else "c2", 64)
pc = ws.PointCloud(ctx, gpc)
for ns in (1, 4):
    rs = [ws.GaussianRenderer(ctx, "rgba32float", 3, False) for _ in range(ns)]
    tg = [torch.empty((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(ns)]
    st = [torch.cuda.current_stream().cuda_stream] + [torch.cuda.Stream().cuda_stream for _ in range(ns - 1)]
    def frame(i):
        k = i % ns
        rs[k].prepare(pc, views[i % 64], stream=st[k]); rs[k].render(pc, target_ptr=tg[k].data_ptr(), stream=st[k])
    for i in range(20): frame(i)
    torch.cuda.synchronize()
    enq, tot = [], []
    for rep in range(10):
        t0 = time.perf_counter()
        for i in range(16): frame(i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        enq.append((t1 - t0) / 16); tot.append((t2 - t0) / 16)
    print(f"streams {ns}: enqueue {1e6*min(enq):.1f} us/frame (median {1e6*sorted(enq)[5]:.1f}), burst total {1e6*min(tot):.1f} us/frame")
    for r in rs: r.close()
