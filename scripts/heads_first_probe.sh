#!/bin/bash
# WS_BATCH_HEADS_FIRST A/B: the driver's form of the bench line (20 steps) and the 1000-step line, hd1m / c2 / c3
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04_${TAG:-heads}
mkdir -p $OUT
R=$OUT/heads_first.txt
: > $R
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -k "heads_first or view_batch" 2>&1 | tail -3 >> $R
for W in hd1m c3; do
  for H in 0 1 0 1 0 1; do
    for K in 20 1000; do
      [[ $K == 1000 && $H$W == *c3 && $K == 1000 ]] && KK=300 || KK=$K
      line=$(WS_BATCH_HEADS_FIRST=$H python bench.py --gpus 1 --steps $KK --warmup 5 --workload $W --no-cpu-baseline --no-secondary 2>/dev/null | tail -1)
      python -c "
import json,sys
j=json.loads(sys.argv[1]); print('$W heads_first=$H steps', j['steps'], 'fps', round(j['value'],1), 'elapsed_ms', round(j['ms_per_step']*j['steps'],3))" "$line" >> $R
    done
  done
done
cat $R
