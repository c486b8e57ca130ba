#!/usr/bin/env python3
"""Stand-alone timing probe of the depth-sort paths (run under rocprofv3 --kernel-trace --stats):
   python scripts/dsort_probe.py N DIST [reps]     DIST = uniform | peaked
WS_DEPTH_SORT selects the path (adaptive / twolevel); 'uniform' draws keys uniformly from a 2^24-wide range above a base,
'peaked' draws float depth keys of a Gaussian blob (most keys in a few top digits)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "web-splat_amd"))
import websplat as ws  # noqa: E402

n = int(sys.argv[1])
dist = sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rng = np.random.default_rng(0)
if dist == "uniform":
    keys = (0x40800000 + rng.integers(0, 1 << 24, size=n)).astype(np.uint32)
else:
    z = np.clip(rng.normal(8.0, 1.2, size=n), 0.5, 15.5).astype(np.float32)
    keys = (np.float32(16.0) - z).view(np.uint32)
ctx = ws.Context(0)
s = ws.GPURSSorter(ctx, n)
dk, dv, da = ctx.malloc(n * 4), ctx.malloc(n * 4), ctx.malloc(n * 4)
idx = np.arange(n, dtype=np.uint32)
for _ in range(reps):
    ctx.upload(dk, keys)
    ctx.upload(dv, idx)
    ctx.upload(da, idx)
    s.sort_depth(dk, dv, n, da)
    ctx.sync()
out = ctx.download(dk, (n,), np.uint32)
assert np.array_equal(out, np.sort(keys)), "not sorted"
print("ok", n, dist, os.environ.get("WS_DEPTH_SORT", "default"))
