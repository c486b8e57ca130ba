import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "web-splat_amd"), os.path.join(ROOT, "tests"), ROOT]
import numpy as np, torch
import websplat as ws, bench
ctx = ws.Context(0)
gpc, views, (w, h), _ = bench.build_workload(ws, "c2", 64)
pc = ws.PointCloud(ctx, gpc)
K = 1000
def run_python(ns, null_first=True):
    rs = [ws.GaussianRenderer(ctx, "rgba32float", 3, False) for _ in range(ns)]
    tg = [torch.empty((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(ns)]
    st = ([torch.cuda.current_stream().cuda_stream] if null_first else [torch.cuda.Stream().cuda_stream]) + [torch.cuda.Stream().cuda_stream for _ in range(ns - 1)]
    def frame(i):
        k = i % ns
        rs[k].prepare(pc, views[i % 64], stream=st[k]); rs[k].render(pc, target_ptr=tg[k].data_ptr(), stream=st[k])
    for i in range(200): frame(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K): frame(i)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    for r in rs: r.close()
    return K / dt
def run_batch(ns):
    b = ws.ViewBatch(ctx, "rgba32float", 3, False, ns)
    tg = [torch.empty((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(ns)]
    vs = [views[i % 64] for i in range(K)]
    arr = ws.ViewBatch.pack_views(vs)
    ptrs = [tg[i % ns].data_ptr() for i in range(K)]
    import ctypes as C
    parr = (C.c_void_p * K)(*ptrs)
    b.render(pc, ws.ViewBatch.pack_views(vs[:200]), ptrs[:200], w * 16); torch.cuda.synchronize()
    t0 = time.perf_counter()
    b.render(pc, arr, parr, w * 16)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    b.close()
    return K / dt, (t1 - t0) / K * 1e6
mode = os.environ.get("AB_MODE", "")
if mode == "torchstreams":
    ss = [torch.cuda.Stream() for _ in range(3)]
    x = torch.zeros(1024, device="cuda")
    for s_ in ss:
        with torch.cuda.stream(s_):
            x.add_(1)
    torch.cuda.synchronize()
elif mode == "nullkernel":
    x = torch.zeros(1024, device="cuda"); x.add_(1); torch.cuda.synchronize()
f, enq = run_batch(4)
print(f"mode={mode} FIRST: batch API, 4 library streams: {f:.1f}  (enqueue {enq:.1f} us/frame)")
f, enq = run_batch(4)
print(f"SECOND: batch API, 4 library streams: {f:.1f}  (enqueue {enq:.1f} us/frame)")
for rep in range(2):
    print("python loop, null + 3 torch streams:", round(run_python(4, True), 1))
    print("python loop, 4 torch streams       :", round(run_python(4, False), 1))
    for ns in (4, 6):
        f, enq = run_batch(ns)
        print(f"batch API, {ns} library streams      : {f:.1f}  (enqueue {enq:.1f} us/frame)")
