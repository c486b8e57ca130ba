#!/bin/bash
# One scripted GPU batch (run through gpurun): tests -> smoke -> bench -> rocprofv3.  Everything lands in gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
STEPS=${STEPS:-200}
WORKLOAD=${WORKLOAD:-c2}
WHAT=${WHAT:-tests,smoke,bench,prof}
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6 > $OUT/device.txt
nproc >> $OUT/device.txt
if [[ $WHAT == *tests* ]]; then
  for f in tests/test_gpu_sort.py tests/test_gpu_preprocess.py tests/test_gpu_render.py; do
    timeout 600 python -m pytest $f -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | tail -60 > $OUT/$(basename $f .py).log
    echo "$f exit=$?" >> $OUT/summary.txt
    tail -3 $OUT/$(basename $f .py).log >> $OUT/summary.txt
  done
fi
if [[ $WHAT == *smoke* ]]; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit=$?" >> $OUT/summary.txt
fi
if [[ $WHAT == *bench* ]]; then
  timeout 900 python bench.py --steps $STEPS --warmup 20 --workload $WORKLOAD > $OUT/bench_$WORKLOAD.json 2> $OUT/bench_$WORKLOAD.err; echo "bench exit=$?" >> $OUT/summary.txt
fi
if [[ $WHAT == *prof* ]]; then
  rm -rf $OUT/prof_$WORKLOAD
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$WORKLOAD -o prof -- python bench.py --steps 50 --warmup 5 --workload $WORKLOAD --no-cpu-baseline > $OUT/prof_$WORKLOAD.log 2>&1; echo "prof exit=$?" >> $OUT/summary.txt
  find $OUT/prof_$WORKLOAD -name "*kernel_stats*" | head -3 >> $OUT/summary.txt
  # keep the traces small: drop the per-dispatch trace, keep the stats
  find $OUT/prof_$WORKLOAD -name "*kernel_trace*" -size +20M -delete
fi
cat $OUT/summary.txt
