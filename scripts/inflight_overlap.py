"""Frames in flight under the tracer: who runs beside whom (round-5 verdict item 1, step 1).
   From a rocprofv3 kernel trace of `bench.py --streams S`:
     (a) hardware queues: per Queue_Id its busy share, and the matrix "share of queue i's busy time during which queue j has a
         kernel executing too" -- two queues that NEVER overlap share a dispatch pipe (or are serialised by the runtime);
     (b) kernel labels: the matrix "us per frame during which a kernel of label X and one of label Y (of another queue) both
         run", the blend's time alone / in company, K1's time beside a blend;
     (c) idle us per frame, concurrency histogram, mean in-flight duration per label.
   usage: python scripts/inflight_overlap.py <kernel_trace.csv> [skip_first_n_frames] [--json out.json]"""
import csv
import json
import re
import sys
from collections import defaultdict

LABEL = re.compile(r"(k_\w+|fillBuffer\w*|copyBuffer\w*)")


def load(path, skip):
    ev = []
    for r in csv.DictReader(open(path)):
        m = LABEL.search(r["Kernel_Name"])
        name = m.group(1) if m else r["Kernel_Name"][:24]
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, str(r.get("Queue_Id", "?"))))
    ev.sort()
    k1 = [i for i, e in enumerate(ev) if e[2] == "k_preprocess"]
    if len(k1) > skip + 16:
        ev = ev[k1[skip]:k1[-8]]
    return ev


def sweep(ev, key_of):
    """-> (alone[key], pair[(a, b)], level[n], wall): ns during which only `a` runs / a and b run together (a <= b) / n kernels run."""
    pts = []
    for s, e, n, q in ev:
        k = key_of(n, q)
        pts.append((s, 1, k))
        pts.append((e, -1, k))
    pts.sort(key=lambda p: (p[0], p[1]))
    active = defaultdict(int)
    busy = defaultdict(float)
    pair = defaultdict(float)
    level = defaultdict(float)
    last = pts[0][0]
    nact = 0
    for t, d, k in pts:
        dt = t - last
        if dt > 0:
            level[nact] += dt
            keys = [a for a, c in active.items() if c > 0]
            for a in keys:
                busy[a] += dt
            for i, a in enumerate(keys):
                for b in keys[i + 1:]:
                    pair[tuple(sorted((a, b)))] += dt
                if active[a] > 1:
                    pair[(a, a)] += dt
        active[k] += d
        nact += d
        last = t
    return busy, pair, level, pts[-1][0] - pts[0][0]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    if out_json in args:
        args.remove(out_json)
    ev = load(args[0], int(args[1]) if len(args) > 1 else 40)
    frames = sum(1 for e in ev if e[2] == "k_preprocess") or 1
    res = {"frames": frames}
    # (c) per label
    dur = defaultdict(list)
    for s, e, n, q in ev:
        dur[n].append((e - s) / 1e3)
    busy_q, pair_q, level, wall = sweep(ev, lambda n, q: q)
    res["us_per_frame"] = wall / 1e3 / frames
    res["frames_per_s_under_tracer"] = frames / (wall / 1e9)
    res["kernel_us_per_frame"] = sum(sum(d) for d in dur.values()) / frames
    res["mean_concurrency"] = res["kernel_us_per_frame"] / res["us_per_frame"]
    res["idle_us_per_frame"] = level.get(0, 0.0) / 1e3 / frames
    res["share_of_wall_with_k_kernels"] = {str(k): round(v / wall, 4) for k, v in sorted(level.items())}
    res["labels"] = {n: {"launches_per_frame": round(len(d) / frames, 2), "mean_us": round(sum(d) / len(d), 2),
                         "us_per_frame": round(sum(d) / frames, 2)} for n, d in sorted(dur.items(), key=lambda kv: -sum(kv[1]))}
    # (a) queues
    qs = sorted(busy_q, key=lambda x: (len(x), x))
    res["queues"] = {q: {"busy_share_of_wall": round(busy_q[q] / wall, 3),
                         "overlap_share_with": {p: round(pair_q.get(tuple(sorted((q, p))), 0.0) / busy_q[q], 3) for p in qs if p != q}}
                     for q in qs}
    res["queue_pairs_that_never_overlap"] = [[a, b] for i, a in enumerate(qs) for b in qs[i + 1:]
                                            if pair_q.get(tuple(sorted((a, b))), 0.0) < 0.01 * min(busy_q[a], busy_q[b])]
    # (b) labels of DIFFERENT frames running together: us per frame
    busy_l, pair_l, _, _ = sweep(ev, lambda n, q: n)
    top = [n for n, _ in sorted(busy_l.items(), key=lambda kv: -kv[1])][:8]
    res["label_overlap_us_per_frame"] = {a: {b: round(pair_l.get(tuple(sorted((a, b))), 0.0) / 1e3 / frames, 2) for b in top} for a in top}
    res["label_busy_us_per_frame"] = {a: round(busy_l[a] / 1e3 / frames, 2) for a in top}
    # the blend: alone, beside another blend, beside K1, beside small kernels only
    pts = []
    for s, e, n, q in ev:
        pts.append((s, 1, n))
        pts.append((e, -1, n))
    pts.sort(key=lambda p: (p[0], p[1]))
    act = defaultdict(int)
    last = pts[0][0]
    cls = defaultdict(float)
    for t, d, n in pts:
        dt = t - last
        if dt > 0:
            nb, nk = act["k_blend"], act["k_preprocess"]
            others = sum(c for k, c in act.items() if k not in ("k_blend", "k_preprocess"))
            if nb == 0 and nk == 0 and others == 0:
                cls["idle"] += dt
            elif nb == 0:
                cls["no_blend_running" + ("_k1" if nk else "_small_only")] += dt
            elif nb >= 2:
                cls["two_or_more_blends"] += dt
            elif nk:
                cls["one_blend_beside_k1"] += dt
            elif others:
                cls["one_blend_beside_small_kernels"] += dt
            else:
                cls["one_blend_alone"] += dt
        act[n] += d
        last = t
    res["wall_share_by_what_runs"] = {k: round(v / wall, 4) for k, v in sorted(cls.items(), key=lambda kv: -kv[1])}
    print(json.dumps(res, indent=1))
    if out_json:
        json.dump(res, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
