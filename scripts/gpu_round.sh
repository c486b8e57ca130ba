#!/bin/bash
# One scripted GPU batch (run through gpurun): tests -> smoke -> PMC traffic -> PMC VALU -> bench -> rocprofv3 stats.
# Everything lands in gpurun_out/; copy what should be judged into profiles/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
STEPS=${STEPS:-400}
WORKLOAD=${WORKLOAD:-c2}
WHAT=${WHAT:-tests,smoke,traffic,valu,bench,prof}
rm -f $OUT/summary.txt
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6 > $OUT/device.txt
nproc >> $OUT/device.txt
if [[ $WHAT == *tests* ]]; then
  timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -30 > $OUT/tests_gpu.log
  echo "tests exit=$?" >> $OUT/summary.txt; tail -2 $OUT/tests_gpu.log >> $OUT/summary.txt
fi
if [[ $WHAT == *smoke* ]]; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit=$?" >> $OUT/summary.txt
fi
if [[ $WHAT == *traffic* ]]; then
  bash scripts/pmc_traffic.sh $WORKLOAD > $OUT/traffic.log 2>&1; echo "traffic exit=$?" >> $OUT/summary.txt
  mkdir -p profiles; cp $OUT/traffic_$WORKLOAD.json profiles/traffic_$WORKLOAD.json 2>/dev/null  # bench.py reads it
fi
if [[ $WHAT == *valu* ]]; then
  rm -rf $OUT/pmc_valu
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_valu -o pmc -- python bench.py --steps 6 --warmup 2 --streams 1 --no-cpu-baseline --workload $WORKLOAD > $OUT/pmc_valu.log 2>&1
  python scripts/pmc_valu.py $OUT/pmc_valu/pmc_counter_collection.csv $OUT/valu_$WORKLOAD.json > $OUT/valu.log 2>&1; echo "valu exit=$?" >> $OUT/summary.txt
  cp $OUT/valu_$WORKLOAD.json profiles/valu_$WORKLOAD.json 2>/dev/null  # bench.py reads it
fi
if [[ $WHAT == *bench* ]]; then
  timeout 900 python bench.py --steps $STEPS --warmup 20 --workload $WORKLOAD > $OUT/bench_$WORKLOAD.json 2> $OUT/bench_$WORKLOAD.err; echo "bench exit=$?" >> $OUT/summary.txt
fi
if [[ $WHAT == *prof* ]]; then
  rm -rf $OUT/prof_$WORKLOAD
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$WORKLOAD -o prof -- python bench.py --steps 50 --warmup 5 --streams 1 --workload $WORKLOAD --no-cpu-baseline > $OUT/prof_$WORKLOAD.log 2>&1; echo "prof exit=$?" >> $OUT/summary.txt
  python scripts/frame_timeline.py $OUT/prof_$WORKLOAD/prof_kernel_trace.csv > $OUT/prof_${WORKLOAD}_timeline.txt 2>&1
  # keep the merge small: drop the per-dispatch trace, keep the stats
  find $OUT/prof_$WORKLOAD -name "*kernel_trace*" -size +8M -delete
fi
cat $OUT/summary.txt
