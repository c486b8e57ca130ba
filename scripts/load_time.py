"""Scene load time: an INRIA PLY of N vertices (248 B each) -> resident scene, with the per-vertex
conversion (io/ply.rs:50-100) on the GPU (default: ws_pointcloud_create_from_ply_rows / k_ply_decode) and on the host
(WS_PLY_DECODE=host: OpenMP ws_ply_rows_convert + ws_pointcloud_create's re-layout).   python scripts/load_time.py [N]"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "web-splat_amd"), ROOT]
import numpy as np  # noqa: E402
import websplat as ws  # noqa: E402
from websplat import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
rows = synth.scene_c3(n=n, seed=2)
ctx = ws.Context(0)
ctx_host = ws.Context(0, ws.config_from_env(ply_decode_host=1))  # (the switch belongs to the context; the library reads no environment)
out = {"vertices": n, "file_MB": round(n * 248 / 1e6, 1), "host_threads": os.cpu_count()}
with tempfile.TemporaryDirectory(dir="/tmp") as td:
    path = os.path.join(td, "scene.ply")
    synth.write_ply(path, rows, 3)
    del rows
    t0 = time.perf_counter()
    raw = np.fromfile(path, dtype=np.uint8)       # the page cache is warm after the write: what reading the bytes costs
    out["read_file_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    del raw
    for mode in ("gpu", "host", "gpu", "host"):
        t0 = time.perf_counter()
        pc = ws.PointCloud.load(ctx_host if mode == "host" else ctx, path)
        ctx.sync()
        out.setdefault(f"load_{mode}_decode_ms", []).append(round((time.perf_counter() - t0) * 1e3, 1))
        pc.close()
ctx_host.close()
ctx.close()
print(json.dumps(out))
