#!/bin/bash
# One submission thread per slot (WS_BATCH_THREADS=1) against the single submission thread on hd1m, now that the host's
# run-ahead is bounded: frames/s and host cores at 1000 steps and in the driver's 20-step form; same box, alternating.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_threads_ab; mkdir -p $OUT; rm -f $OUT/summary.txt
val() { python -c "
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=j['config']
print(round(j['value'],1), 'busy', round(c.get('host_cores_busy_per_rank',0),2), 'threads', [(t['thread'],t['cores']) for t in (c.get('host_threads') or [])[:6]])" $1 2>&1; }
for k in 1 2 3; do
  for S in 1000 20; do
    for V in "one=WS_BATCH_THREADS=0" "perslot=WS_BATCH_THREADS=1" "perslot_q3=WS_BATCH_THREADS=1,WS_BATCH_QUEUE_DEPTH=3"; do
      N=${V%%=*}; E=${V#*=}
      env ${E//,/ } timeout 300 python bench.py --gpus 1 --steps $S --warmup $([[ $S == 20 ]] && echo 5 || echo 50) --no-secondary --no-cpu-baseline ${W:+--workload $W} > $OUT/${N}_${S}_$k.json 2> $OUT/${N}_${S}_$k.err
      echo "$N steps$S run$k: $(val $OUT/${N}_${S}_$k.json)" >> $OUT/summary.txt
    done
  done
done
cat $OUT/summary.txt
