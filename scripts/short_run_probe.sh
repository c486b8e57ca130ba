#!/bin/bash
# The driver times `bench.py --steps 20 --warmup 5`: a 3-ms region in which the fill and drain of the frames in flight are a
# visible share.  Elapsed time against K for 2..5 frames in flight, and what the closing bracket (sync + barrier) costs.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04_${TAG:-short}
mkdir -p $OUT
R=$OUT/short_run.txt
: > $R
for S in 4 3 2 5; do
  for K in 20 20 20 40 80 160; do
    line=$(WS_BENCH_DEBUG=1 python bench.py --steps $K --warmup 5 --streams $S --no-cpu-baseline --no-secondary 2> $OUT/err.txt | tail -1)
    dbg=$(grep "bench debug" $OUT/err.txt | tail -1)
    python - "$S" "$K" "$line" "$dbg" >> $R <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[3])
    print(f"streams {sys.argv[1]} steps {int(sys.argv[2]):4d}  fps {j['value']:8.1f}  elapsed_ms {j['ms_per_step'] * j['steps']:.3f}  {sys.argv[4]}")
except Exception as e:
    print("FAILED", sys.argv[1], sys.argv[2], e)
PY
  done
done
for extra in "--no-dist" ; do
  for K in 20 20 20; do
    line=$(WS_BENCH_DEBUG=1 python bench.py --steps $K --warmup 5 --streams 4 $extra --no-cpu-baseline --no-secondary 2> $OUT/err.txt | tail -1)
    dbg=$(grep "bench debug" $OUT/err.txt | tail -1)
    python -c "
import json,sys
j=json.loads(sys.argv[1]); print('streams 4 steps', j['steps'], '$extra', 'fps', round(j['value'],1), sys.argv[2])" "$line" "$dbg" >> $R
  done
done
cat $R
