#!/bin/bash
# VALU / wait accounting of the binning kernels, default against WS_BIN_SHIFT=1 (why is k_bin_emit slower with half the entries?)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_pmc_emit; mkdir -p $OUT
for V in 0 1; do
  rm -rf $OUT/v$V
  WS_BIN_SHIFT=$V timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/v$V -o pmc -- python bench.py --steps 6 --warmup 2 --streams 1 --no-cpu-baseline --no-dist --workload hd1m > $OUT/v$V.log 2>&1
  python - $OUT/v$V/pmc_counter_collection.csv $V <<'PY'
import csv,sys
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"]
    for k in ("k_bin_emit","k_bin_prefix","k_blend","k_preprocess"):
        if k in n: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in acc.items():
    m={a:sum(b)/len(b) for a,b in c.items()}
    cyc=m.get("GRBM_GUI_ACTIVE",0)/8
    print("bin_shift",sys.argv[2],k,"VALU insts",int(m.get("SQ_INSTS_VALU",0)),"kernel cycles",int(cyc),"wave cycles",int(m.get("SQ_WAVE_CYCLES",0)),"wait_any",round(m.get("SQ_WAIT_ANY",0)/max(m.get("SQ_WAVE_CYCLES",1),1),2),"wait_inst",round(m.get("SQ_WAIT_INST_ANY",0)/max(m.get("SQ_WAVE_CYCLES",1),1),2))
PY
  find $OUT/v$V -size +2M -delete
done
