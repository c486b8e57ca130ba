"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into HBM bytes per launch per kernel label.

  python scripts/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

Units and corrections (MI355X_MICROARCH.md, section HBM): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of a coalesced streaming read, so it is doubled.  Calibration on this code base:
k_sort_tile_hist reads exactly 4*D bytes and reports 0.50 of them (dword loads, 256 B per wave instruction);
k_blend writes exactly W*H*16 bytes and WRITE_SIZE reports 1.00 of them; kernels that write many short runs
(radix scatter, emit) report ~1.2x their algorithmic write bytes (partial lines).
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
    if "k_blend" in name:
        return "k_blend"
    if "k_dsort_hist" in name:
        return "depth:k_dsort_hist"
    if "k_dsort_scatter" in name:
        return "depth:k_dsort_scatter"
    m = re.search(r"k_sort_scatter<(?:false|true), (\d+), (false|true), (\d+)(?:, (false|true))?>", name)
    if m:  # <LOOKBACK, KPT, RANGES, BITS, CARRY>: the depth sort carries the tile rectangles (CARRY); the tile-id sort does not
        carry = m.group(4) == "true"
        kpt, bits = int(m.group(1)), int(m.group(3))
        which = "depth" if carry or (bits == 8 and m.group(2) == "false" and kpt == 4) else "tiles"
        return f"{which}:k_sort_scatter"
    m = re.search(r"k_sort_tile_hist<(\d+)>", name)
    if m:
        return ("depth" if int(m.group(1)) == 4 else "sort") + ":k_sort_tile_hist"
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


if __name__ == "__main__":
    fetch = per_label(sys.argv[1], "FETCH_SIZE")
    write = per_label(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        out[k] = 2.0 * fetch.get(k, 0.0) * 1024.0 + write.get(k, 0.0) * 1024.0
    detail = {k: {"FETCH_SIZE_KiB": fetch.get(k), "WRITE_SIZE_KiB": write.get(k)} for k in out}
    json.dump({**out, "_raw": detail, "_formula": "2*FETCH_SIZE*1024 + WRITE_SIZE*1024 bytes per launch"}, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out))
