#!/bin/bash
# submission threads of a view batch (WS_BATCH_THREADS): tests, then bench lines for small and large scenes, threads off / default
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04_${TAG:-mt}
mkdir -p $OUT
R=$OUT/mt_bench.txt
: > $R
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_drivers.py tests/test_gpu_bench.py -m gpu -q -p no:cacheprovider -k "view_batch or measure or contract or sliver or two_ranks" 2>&1 | tail -6 >> $R
for a in "10000 800 600" "250000 800 600"; do python scripts/mt_enqueue_probe.py $a 2>/dev/null | grep "^N" >> $R; done
for W in c1 hd1m c2; do
  for T in 0 -1 0 -1; do
    line=$(WS_BATCH_THREADS=$T timeout 600 python bench.py --steps 2000 --warmup 50 --workload $W --no-cpu-baseline --no-secondary 2>/dev/null | tail -1)
    python -c "
import json,sys
try:
    j=json.loads(sys.argv[1]); c=j['config']
    print('$W threads=$T fps', round(j['value'],1), 'single', round(c['single_stream_fps'],1), 'enq_ms', round(c['host_enqueue_ms_per_frame'],4), 'cores_busy', round(c['host_cores_busy_per_rank'],2), 'host_bound', c['host_bound'])
except Exception as e:
    print('$W threads=$T FAILED', e)" "$line" >> $R
  done
done
for T in 0 -1; do
  line=$(WS_BATCH_THREADS=$T python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1)
  python -c "
import json,sys
j=json.loads(sys.argv[1]); print('hd1m 20 steps threads=$T fps', round(j['value'],1))" "$line" >> $R
done
cat $R
