"""Frames in flight, measured ON THE DEVICE (round 6; verdict r05 item 1, step 1).

rocprofv3's kernel trace serialises the hardware queues (profiles/r06/inflight_overlap_hd1m_s*.json: never two blends at once,
270 us per frame instead of 137), so the question "do the VALU-bound blend and the HBM-bound K1 overlap or take turns?" cannot be
answered from it.  Here K1 and the compositing kernel of every frame of a view batch leave {first workgroup start, last workgroup
end} on the device's 100-MHz clock (ws_renderer_enable_frame_trace: two 64-bit atomics per workgroup), with NO tracer attached:
  * in-flight duration of K1 and of the blend (against their lone durations),
  * us per frame during which 0 / 1 / 2+ blends run, K1 runs beside a blend, K1 runs beside K1, neither runs (only the small
    dependent kernels -- sorts, binning -- or nothing),
  * per slot: the frame period, and the share of it the slot's own K1 + blend cover.
usage: python scripts/inflight_device_trace.py [workload] [streams...]   -> one JSON object per stream count
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-splat_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import websplat as ws  # noqa: E402


def union_len(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0.0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def sweep(k1, bl, t0, t1):
    """time (us) by (number of K1 running, number of blends running) inside [t0, t1]"""
    pts = []
    for s, e in k1:
        pts += [(s, 0, 1), (e, 0, -1)]
    for s, e in bl:
        pts += [(s, 1, 1), (e, 1, -1)]
    pts.sort()
    n = [0, 0]
    last = t0
    acc = {}
    for t, which, d in pts:
        tt = min(max(t, t0), t1)
        if tt > last:
            key = (min(n[0], 2), min(n[1], 2))
            acc[key] = acc.get(key, 0.0) + (tt - last)
            last = tt
        n[which] += d
    if t1 > last:
        acc[(0, 0)] = acc.get((0, 0), 0.0) + (t1 - last)
    return acc


def run(workload, nstreams, frames=400):
    ctx = ws.Context(0)
    gpc, views, viewport, _ = bench.build_workload(ws, workload, 64)
    w, h = viewport
    pc = ws.PointCloud(ctx, gpc)
    batch = ws.ViewBatch(ctx, "rgba32float", gpc.sh_deg, gpc.compressed, nstreams)
    targets = [torch.empty((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(nstreams)]
    packed = ws.ViewBatch.pack_views(views)
    import ctypes as C

    def plan(first, count):
        arr = (type(packed[0]) * count)()
        ptrs = (C.c_void_p * count)()
        for j in range(count):
            arr[j] = packed[(first + j) % len(views)]
            ptrs[j] = targets[(first + j) % nstreams].data_ptr()
        return arr, ptrs
    pitch = w * batch.texel_bytes
    warm = plan(0, 8 * nstreams)
    batch.render(pc, warm[0], warm[1], pitch)
    torch.cuda.synchronize()
    per_slot = (frames + nstreams - 1) // nstreams
    for s in range(nstreams):
        batch.renderer(s).enable_frame_trace(per_slot)
    timed = plan(8 * nstreams, frames)
    import time
    t0 = time.perf_counter()
    batch.render(pc, timed[0], timed[1], pitch)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    err = batch.errors()
    tr = [batch.renderer(s).frame_trace().astype(np.float64) / 100.0 for s in range(nstreams)]   # us
    for s in range(nstreams):
        batch.renderer(s).enable_frame_trace(0)
    batch.close()
    pc.close()
    ctx.close()
    # steady state: drop the first and last 10 % of every slot's frames
    k1, bl, periods, own = [], [], [], []
    for t in tr:
        n = len(t)
        lo, hi = n // 10, n - n // 10
        k1 += [(a, b) for a, b, _, _ in t[lo:hi]]
        bl += [(c, d) for _, _, c, d in t[lo:hi]]
        periods.append(float(np.mean(np.diff(t[lo:hi, 0]))))
        own.append(float(np.mean((t[lo:hi, 1] - t[lo:hi, 0]) + (t[lo:hi, 3] - t[lo:hi, 2])) / periods[-1]))
    a = max(min(s for s, _ in k1), min(s for s, _ in bl))
    b = min(max(e for _, e in k1), max(e for _, e in bl))
    acc = sweep(k1, bl, a, b)
    nframes = sum(1 for s, e in k1 if a <= s and e <= b)
    span = b - a
    per_frame = span / max(nframes, 1)

    def share(pred):
        return sum(v for k, v in acc.items() if pred(*k)) / span * per_frame
    out = {"workload": workload, "frames_in_flight": nstreams, "frames": frames, "error_bits": err,
           "frames_per_s_untraced_clock": frames / elapsed, "us_per_frame": per_frame,
           "k1_in_flight_us": float(np.mean([e - s for s, e in k1])), "blend_in_flight_us": float(np.mean([e - s for s, e in bl])),
           "us_per_frame_with": {
               "no K1, no blend (small kernels or idle)": share(lambda k, bb: k == 0 and bb == 0),
               "K1 only": share(lambda k, bb: k >= 1 and bb == 0),
               "one blend, no K1": share(lambda k, bb: k == 0 and bb == 1),
               "one blend beside K1": share(lambda k, bb: k >= 1 and bb == 1),
               "two or more blends, no K1": share(lambda k, bb: k == 0 and bb >= 2),
               "two or more blends beside K1": share(lambda k, bb: k >= 1 and bb >= 2),
               "two or more K1": share(lambda k, bb: k >= 2)},
           "union_us_per_frame": {"K1": union_len(k1) / max(nframes, 1), "blend": union_len(bl) / max(nframes, 1),
                                  "K1 or blend": union_len(k1 + bl) / max(nframes, 1)},
           "per_slot_frame_period_us": periods, "per_slot_share_of_period_in_own_k1_and_blend": own}
    return out


if __name__ == "__main__":
    wl = sys.argv[1] if len(sys.argv) > 1 else "hd1m"
    streams = [int(x) for x in sys.argv[2:]] or [1, 4, 5, 6, 8]
    for s in streams:
        print(json.dumps(run(wl, s)))
        sys.stdout.flush()
