"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into HBM bytes per launch per kernel label.

  python scripts/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

Units and corrections.  FETCH_SIZE / WRITE_SIZE are in KiB.  What they count on gfx950 was CALIBRATED on kernels that move
a known number of bytes (scripts/ubench/fetch_calib.hip, profiles/fetch_calibration.json, round 3):
  * coalesced reads, 16 B or 4 B per lane alike: FETCH_SIZE reports exactly HALF of the bytes  -> read factor 2.0
    (MI355X_MICROARCH.md, section HBM: 128-B requests tallied at 64 B);
  * random gathers (4-B words, or the blend's 20-B Splat records): FETCH_SIZE is the number of 64-B lines requested x 64 B,
    1.00 .. 1.10 of the lines the gathers touch -> read factor 1.0; the USEFUL bytes are 6 % (4 B) .. 28 % (20 B) of that;
  * WRITE_SIZE is exact for coalesced stores and counts whole 32-B sectors for scattered ones (8 x the useful bytes of a
    random 4-B store): it is the traffic, factor 1.0.
Rounds 1-2 applied the factor 2 to every kernel; that doubled the blend's gather traffic (c3: "4.0 x the algorithmic
bytes" was 2.4 x).  Each kernel label now carries its class and factor; a kernel that mixes a coalesced index stream with
gathers (k_blend: 4 B of entry index per 20-B record) is priced as gathers, and the coalesced part it under-counts is
bounded by `bytes_if_all_streaming`.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def label_of(name):
    if "k_preprocess" in name:
        return "k_preprocess"
    if "k_bin_prefix" in name:
        return "k_bin_prefix"
    if "k_bin_emit" in name:
        return "k_bin_emit"
    if "k_blend_order" in name:
        return "k_blend_order"
    if "k_blend" in name:
        return "k_blend"
    m = re.search(r"k_sort_scatter<(\d+), (false|true), (\d+), (false|true), (false|true)>", name)
    if m:  # <KPT, RANGES, BITS, CARRY, KEY16>: the depth sort carries the footprint words (CARRY) with 32-bit keys;
        # the tile-id sort has 16-bit keys and its last pass records the tile ranges
        carry, key16, ranges = m.group(4) == "true", m.group(5) == "true", m.group(2) == "true"
        which = "tiles" if (key16 or ranges) and not carry else "depth"
        return f"{which}:k_sort_scatter"
    if "k_sort_col_scan" in name:
        return "k_sort_col_scan"
    m = re.search(r"k_sort_tile_hist<(\d+), (false|true)>", name)
    if m:  # <KPT, KEY16>: 16-bit keys are the tile-id sort's
        return ("tiles" if m.group(2) == "true" else "depth") + ":k_sort_tile_hist"
    return None


def per_label(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        lab = label_of(r["Kernel_Name"])
        if lab:
            acc[lab].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


# read factor by access pattern of the kernel's dominant fetch stream (profiles/fetch_calibration.json)
GATHER_KERNELS = ("k_blend",)            # random 20-B record gathers dominate the fetches
READ_FACTOR = {"stream": 2.0, "gather": 1.0}


if __name__ == "__main__":
    fetch = per_label(sys.argv[1], "FETCH_SIZE")
    write = per_label(sys.argv[2], "WRITE_SIZE")
    out, detail = {}, {}
    for k in sorted(set(fetch) | set(write)):
        cls = "gather" if k in GATHER_KERNELS else "stream"
        f, w = fetch.get(k, 0.0) * 1024.0, write.get(k, 0.0) * 1024.0
        out[k] = READ_FACTOR[cls] * f + w
        detail[k] = {"FETCH_SIZE_bytes_raw": f, "WRITE_SIZE_bytes_raw": w, "class": cls, "read_factor": READ_FACTOR[cls],
                     "write_factor": 1.0, "bytes": out[k], "bytes_if_all_streaming": 2.0 * f + w}
    json.dump({**out, "_detail": detail,
               "_formula": "read_factor * FETCH_SIZE + WRITE_SIZE (KiB -> bytes) per launch; read_factor 2.0 for coalesced "
                           "streams, 1.0 for random gathers (calibrated: profiles/fetch_calibration.json)"},
              open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out))
