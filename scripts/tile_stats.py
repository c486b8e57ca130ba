"""Analysis helper (GPU): per-tile list length / consumed-entry distribution and per-kernel times of one frame."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "web-splat_amd"), os.path.join(ROOT, "tests"), ROOT]
import numpy as np
import websplat as ws
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
ctx = ws.Context(0)
gpc, views, (w, h), _ = bench.build_workload(ws, name, 8)
pc = ws.PointCloud(ctx, gpc)
r = ws.GaussianRenderer(ctx, "rgba32float", 3, False)
r.enable_capture(True)
r.enable_timers(2)
for vi in (0, 3):
    for _ in range(3):
        r.prepare(pc, views[vi]); r.render(pc)
    kt = r.kernel_times()
    st = r.frame_stats()
    ts = r.tile_stats(with_consumed=True)
    ll, co = ts["list_len"].astype(np.int64), ts["consumed"].astype(np.int64)
    print(f"view {vi}: V={st['num_visible']} D={st['num_tile_entries']} tiles={ll.size}")
    print("  list_len  pct50/90/99/max:", [int(np.percentile(ll, p)) for p in (50, 90, 99, 100)], "sum", int(ll.sum()))
    print("  consumed  pct50/90/99/max:", [int(np.percentile(co, p)) for p in (50, 90, 99, 100)], "sum", int(co.sum()))
    print("  tiles with consumed > 1024:", int((co > 1024).sum()), " > 2048:", int((co > 2048).sum()), " > 4096:", int((co > 4096).sum()))
    ws_ = r.wave_stats().astype(np.int64)
    tw, th = ctx.tile_size()
    nw = (tw // 8) * (th // 8)
    per_wave, lock = ws_[:, :nw], ws_[:, 16]
    print(f"  records composited: sum over waves {int(per_wave.sum())}; per tile: mean-wave sum {per_wave.mean(1).sum():.0f}, "
          f"max-wave sum {int(per_wave.max(1).sum())}, lock-step (sum over batches of the batch maximum) {int(lock.sum())}"
          f"  -> lock-step / mean = {lock.sum() / max(per_wave.mean(1).sum(), 1):.2f}, max / mean = {per_wave.max(1).sum() / max(per_wave.mean(1).sum(), 1):.2f}")
    print("  kernel times (us):", " ".join(f"{n}={ms*1e3:.1f}" for n, ms in kt), " total=%.1f" % (sum(ms for _, ms in kt) * 1e3))
r.close(); pc.close(); ctx.close()
