"""With N frames in flight (one stream each): how long does a frame occupy its stream, and does the stream ever wait for
the host?  Every frame is bracketed by two timing events on its stream; the start event of frame i+N sits right behind
the end event of frame i, so start(i+N) - end(i) > 0 means the stream ran dry (the host had not submitted yet).
usage: python scripts/stream_gaps.py [workload] [streams...]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path[:0] = ["web-splat_amd", "tests", "."]
import numpy as np, torch
import websplat as ws, bench
ctx = ws.Context(0)
wl = sys.argv[1] if len(sys.argv) > 1 else "hd1m"
gpc, views, (w, h), _ = bench.build_workload(ws, wl, 64)
pc = ws.PointCloud(ctx, gpc)
for ns in [int(x) for x in sys.argv[2:]] or [1, 2, 4]:
    rs = [ws.GaussianRenderer(ctx, "rgba32float", 3, False) for _ in range(ns)]
    tg = [torch.empty((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(ns)]
    ts = [torch.cuda.Stream() for _ in range(ns)]
    st = [s.cuda_stream for s in ts]
    K = 400
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    EVENTS = os.environ.get("GAPS_EVENTS", "1") != "0"
    hp, hr = [], []
    def frame(i, timed):
        k = i % ns
        if timed and EVENTS: ev[i][0].record(ts[k])
        a = time.perf_counter()
        rs[k].prepare(pc, views[i % 64], stream=st[k])
        b = time.perf_counter()
        rs[k].render(pc, target_ptr=tg[k].data_ptr(), stream=st[k])
        c = time.perf_counter()
        if timed: hp.append(b - a); hr.append(c - b)
        if timed and EVENTS: ev[i][1].record(ts[k])
    for i in range(40): frame(i, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K): frame(i, True)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    hp_, hr_ = np.array(hp[40:]) * 1e6, np.array(hr[40:]) * 1e6
    print(f"   host: prepare() mean {hp_.mean():.1f} median {np.median(hp_):.1f} p90 {np.percentile(hp_, 90):.1f} max {hp_.max():.0f} us; "
          f"render() mean {hr_.mean():.1f} median {np.median(hr_):.1f} p90 {np.percentile(hr_, 90):.1f} max {hr_.max():.0f} us")
    if not EVENTS:
        print(f"{wl} streams {ns} (no events): {1e6 * t_all / K:.1f} us/frame (host enqueue {1e6 * t_enq / K:.1f})")
        for r in rs: r.close()
        continue
    dur = np.array([ev[i][0].elapsed_time(ev[i][1]) for i in range(40, K)]) * 1e3
    gap = np.array([ev[i][1].elapsed_time(ev[i + ns][0]) for i in range(40, K - ns)]) * 1e3
    print(f"{wl} streams {ns}: {1e6 * t_all / K:.1f} us/frame (host enqueue {1e6 * t_enq / K:.1f}); frame on its stream: mean {dur.mean():.1f} "
          f"median {np.median(dur):.1f} us; stream idle before the next frame: mean {gap.mean():.1f} median {np.median(gap):.1f} us "
          f"(> 5 us in {100 * (gap > 5).mean():.0f} % of the frames)")
    for r in rs: r.close()
