"""VALU issue accounting per kernel label from one rocprofv3 --pmc pass (SQ_INSTS_VALU, SQ_WAVE_CYCLES,
SQ_WAIT_ANY, SQ_WAIT_INST_ANY, GRBM_GUI_ACTIVE) of bench.py --streams 1.

  python scripts/pmc_valu.py <counter_collection.csv> <out.json>

issue_util = SQ_INSTS_VALU x 2.4 cycles / (1024 SIMDs x kernel cycles): the share of the chip's wave64 VALU issue
slots the kernel filled (2.4 cycles per wave64 instruction per SIMD is what scripts/ubench/valu_rate.hip measures on
MI355X; kernel cycles = GRBM_GUI_ACTIVE / 8, the counter is summed over the 8 XCDs).  wave_wait / wave_stall =
SQ_WAIT_ANY / SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES: the share of the resident waves' time parked at s_waitcnt or a
barrier / stalled at issue (MI355X_MICROARCH.md, SQ counters)."""
import csv
import json
import sys
from collections import defaultdict

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_traffic import label_of  # noqa: E402  (same kernel-name -> label mapping)

acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    lab = label_of(r["Kernel_Name"])
    if lab:
        acc[lab][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for lab, c in sorted(acc.items()):
    m = {k: sum(v) / len(v) for k, v in c.items()}
    cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if not cyc or "SQ_INSTS_VALU" not in m:
        continue
    out[lab] = {"valu_insts_per_launch": m["SQ_INSTS_VALU"], "kernel_cycles": cyc,
                "issue_util": m["SQ_INSTS_VALU"] * 2.4 / (1024.0 * cyc),
                "wave_wait": m.get("SQ_WAIT_ANY", 0.0) / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else None,
                "wave_stall": m.get("SQ_WAIT_INST_ANY", 0.0) / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else None}
json.dump({**out, "_formula": "issue_util = SQ_INSTS_VALU * 2.4 / (1024 * GRBM_GUI_ACTIVE / 8)"}, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
