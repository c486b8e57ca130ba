cd "${GRAFT_REPO_ROOT:-/root/repo}"
for c in 1 2 3 4 0; do
python - <<PY
import os, sys, time
os.environ["WS_DEBUG_CUT"] = "$c"
sys.path[:0] = ["web-splat_amd", "tests", "."]
import torch, websplat as ws, bench
ctx = ws.Context(0)
gpc, views, (w, h), _ = bench.build_workload(ws, "${WORKLOAD:-c2}", 64)
pc = ws.PointCloud(ctx, gpc)
out = []
for ns in (1, 4):
    rs = [ws.GaussianRenderer(ctx, "rgba32float", 3, False) for _ in range(ns)]
    tg = [torch.empty((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(ns)]
    st = [torch.cuda.current_stream().cuda_stream] + [torch.cuda.Stream().cuda_stream for _ in range(ns - 1)]
    def frame(i):
        k = i % ns
        rs[k].prepare(pc, views[i % 64], stream=st[k]); rs[k].render(pc, target_ptr=tg[k].data_ptr(), stream=st[k])
    for i in range(20): frame(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(400): frame(i)
    torch.cuda.synchronize(); out.append(1e6 * (time.perf_counter() - t0) / 400)
    for r in rs: r.close()
print("cut $c: us/frame  1 stream %.1f   4 streams %.1f" % tuple(out))
PY
done
