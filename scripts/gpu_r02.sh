#!/bin/bash
# Round-2 GPU batch (run through gpurun): tests -> per-workload bench / rocprofv3 kernel stats / PMC traffic.
#   WHAT=tests,bench,prof,traffic  WORKLOADS="hd1m c3"  TAG=v1  STEPS=1000
# Everything lands in gpurun_out/r02_<TAG>/; copy what should be judged into profiles/r02/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-v1}
OUT=gpurun_out/r02_$TAG
mkdir -p $OUT
STEPS=${STEPS:-1000}
WORKLOADS=${WORKLOADS:-hd1m c3}
WHAT=${WHAT:-tests,bench,prof,traffic}
rm -f $OUT/summary.txt
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6 > $OUT/device.txt
nproc >> $OUT/device.txt
if [[ $WHAT == *tests* ]]; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x ${PYTEST_ARGS:-} 2>&1 | tail -40 > $OUT/tests_gpu.log
  echo "tests exit=${PIPESTATUS[0]}" >> $OUT/summary.txt; tail -3 $OUT/tests_gpu.log >> $OUT/summary.txt
  cp gpurun_out/parity_fullsize.json $OUT/ 2>/dev/null
fi
for W in $WORKLOADS; do
  if [[ $WHAT == *bench* ]]; then
    timeout 900 python bench.py --steps $STEPS --warmup 50 --workload $W > $OUT/bench_$W.json 2> $OUT/bench_$W.err; echo "bench $W exit=$?" >> $OUT/summary.txt
  fi
  if [[ $WHAT == *prof* ]]; then
    rm -rf $OUT/prof_$W
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$W -o prof -- python bench.py --steps 60 --warmup 10 --streams 1 --workload $W --no-cpu-baseline --no-dist > $OUT/prof_$W.log 2>&1; echo "prof $W exit=$?" >> $OUT/summary.txt
    python scripts/frame_timeline.py $OUT/prof_$W/prof_kernel_trace.csv > $OUT/prof_${W}_timeline.txt 2>&1
    cp $OUT/prof_$W/prof_kernel_stats.csv $OUT/${W}_kernel_stats.csv 2>/dev/null
    find $OUT/prof_$W -name "*kernel_trace*" -size +4M -delete
  fi
  if [[ $WHAT == *traffic* ]]; then
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf $OUT/pmc_${W}_$c
      timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${W}_$c -o pmc -- python bench.py --steps 6 --warmup 2 --streams 1 --no-cpu-baseline --no-dist --workload $W > $OUT/pmc_${W}_$c.log 2>&1
    done
    python scripts/pmc_traffic.py $OUT/pmc_${W}_FETCH_SIZE/pmc_counter_collection.csv $OUT/pmc_${W}_WRITE_SIZE/pmc_counter_collection.csv $OUT/traffic_$W.json > $OUT/traffic_$W.log 2>&1; echo "traffic $W exit=$?" >> $OUT/summary.txt
    find $OUT/pmc_${W}_FETCH_SIZE $OUT/pmc_${W}_WRITE_SIZE -size +2M -delete 2>/dev/null
  fi
done
cat $OUT/summary.txt
