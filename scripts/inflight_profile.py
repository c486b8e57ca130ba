"""Overlap analysis of a rocprofv3 kernel trace taken with several frames in flight:
   per kernel label: mean duration (to compare with the one-frame-in-flight trace), share of the wall time during which
   k kernels are executing, time with the blend running alone / with company, idle time.
   usage: python scripts/inflight_profile.py <kernel_trace.csv> [skip_first_n_frames]"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ev = []
for r in rows:
    m = re.search(r"(k_\w+|fillBuffer\w*|copyBuffer\w*)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:20]
    if name == "k_sort_scatter":
        name += "<carry>" if "true>(" in r["Kernel_Name"].replace(" ", "") and ", 8, true" in r["Kernel_Name"] else ""
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r["Queue_Id"]))
ev.sort()
k1 = [i for i, e in enumerate(ev) if e[2] == "k_preprocess"]
ev = ev[k1[skip]:k1[-8]]
t0, t1 = ev[0][0], max(e[1] for e in ev)
frames = sum(1 for e in ev if e[2] == "k_preprocess")
wall = (t1 - t0) / 1e3
print(f"{frames} frames in {wall:.0f} us: {wall / frames:.1f} us/frame, queues used: {sorted(set(e[3] for e in ev))}")
dur = defaultdict(list)
for s, e, n, q in ev:
    dur[n].append((e - s) / 1e3)
print("kernel                     launches/frame  mean us   sum us/frame")
tot = 0.0
for n, d in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {n:26s} {len(d) / frames:6.2f}      {sum(d) / len(d):8.1f}  {sum(d) / frames:8.1f}")
    tot += sum(d) / frames
print(f"  sum of kernel durations per frame: {tot:.1f} us  (average concurrency {tot / (wall / frames):.2f})")
# sweep
pts = []
for s, e, n, q in ev:
    pts.append((s, 1, n))
    pts.append((e, -1, n))
pts.sort()
active = defaultdict(int)
level = defaultdict(float)
blend_alone = blend_company = 0.0
last = pts[0][0]
nact = 0
for t, d, n in pts:
    dt = (t - last) / 1e3
    level[nact] += dt
    if active["k_blend"] > 0:
        if nact - active["k_blend"] == 0:
            blend_alone += dt
        else:
            blend_company += dt
    active[n] += d
    nact += d
    last = t
print("share of wall time with k kernels executing:", {k: round(v / wall, 3) for k, v in sorted(level.items())})
print(f"blend executing: alone {blend_alone / wall:.3f}, together with other kernels {blend_company / wall:.3f} of the wall time")
