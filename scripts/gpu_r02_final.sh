#!/bin/bash
# Final round-2 evidence batch: tests, then for every workload bench (default flags: 4 frames in flight, CPU baseline),
# rocprofv3 kernel stats + one-frame timeline, PMC traffic, PMC VALU accounting; scene load time; host submission time.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-final}
OUT=gpurun_out/r02_$TAG
mkdir -p $OUT
TAG=$TAG WHAT=tests bash scripts/gpu_r02.sh
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit=$?" >> $OUT/summary.txt
TAG=$TAG WHAT=bench,prof,traffic WORKLOADS="hd1m c3 c2" bash scripts/gpu_r02.sh
TAG=$TAG WHAT=bench WORKLOADS="c4 c5" STEPS=600 bash scripts/gpu_r02.sh
for W in hd1m c3 c2; do
  rm -rf $OUT/pmc_valu_$W
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_valu_$W -o pmc -- python bench.py --steps 6 --warmup 2 --streams 1 --no-cpu-baseline --no-dist --workload $W > $OUT/pmc_valu_$W.log 2>&1
  python scripts/pmc_valu.py $OUT/pmc_valu_$W/pmc_counter_collection.csv $OUT/valu_$W.json > /dev/null 2>&1
  find $OUT/pmc_valu_$W -size +2M -delete
done
timeout 900 python scripts/load_time.py 5000000 > $OUT/load_time_5m.json 2> $OUT/load_time_5m.err; echo "load_time exit=$?" >> $OUT/summary.txt
timeout 600 python scripts/host_time.py hd1m > $OUT/host_time_hd1m.txt 2>&1; echo "host_time exit=$?" >> $OUT/summary.txt
cat $OUT/summary.txt
