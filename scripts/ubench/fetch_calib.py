#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE against known byte counts (scripts/ubench/fetch_calib.hip) -> calibration factors.

  python scripts/ubench/fetch_calib.py <known.json> <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

factor = bytes the kernel demonstrably moves / bytes the counter reports (KiB * 1024).  For the gathers two denominators
are given: the useful bytes (20 B or 4 B per gather) and the bytes of the 64-B / 128-B lines those gathers touch (a random
20-B record at 4-byte alignment straddles a 64-B line with probability 16/64 and a 128-B line with 16/128)."""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "k_calib_" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].split("::")[-1].split()[-1]
            acc[name].append(float(r["Counter_Value"]) * 1024.0)
    return {k: sum(v[1:]) / max(len(v) - 1, 1) if len(v) > 1 else v[0] for k, v in acc.items()}  # (first launch: cold)


def main():
    known = json.load(open(sys.argv[1]))
    fetch, write = per_kernel(sys.argv[2], "FETCH_SIZE"), per_kernel(sys.argv[3], "WRITE_SIZE")
    out = {}
    for k, kn in known.items():
        e = {"counter_FETCH_bytes": fetch.get(k), "counter_WRITE_bytes": write.get(k), **kn}
        f, w = fetch.get(k) or 0.0, write.get(k) or 0.0
        idx = kn.get("index_read", 0)
        if "read" in kn and f:
            if "gathers" in kn:
                g = kn["gathers"]
                rec = kn["read"] // g
                # the index stream is a coalesced dword read: priced with the streaming-dword factor below
                e["useful_bytes"] = kn["read"]
                e["lines64_bytes"] = g * 64 * (1.0 + (rec - 4) / 64.0)
                e["lines128_bytes"] = g * 128 * (1.0 + (rec - 4) / 128.0)
            else:
                e["factor_read"] = kn["read"] / f
        if "write" in kn and w:
            e["factor_write"] = kn["write"] / w
        out[k] = e
    f4 = out.get("k_calib_read4", {}).get("factor_read")
    for k in ("k_calib_gather20", "k_calib_gather4"):
        e = out.get(k)
        if e and e.get("counter_FETCH_bytes") and f4:
            table_counter = e["counter_FETCH_bytes"] - e["index_read"] / f4   # what the counter shows for the gathers alone
            e["counter_bytes_for_the_gathers"] = table_counter
            e["factor_useful"] = e["useful_bytes"] / table_counter
            e["factor_lines64"] = e["lines64_bytes"] / table_counter
            e["factor_lines128"] = e["lines128_bytes"] / table_counter
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    for k, e in out.items():
        print(k, {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in e.items() if kk.startswith("factor")})


if __name__ == "__main__":
    main()
