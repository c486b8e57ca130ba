#!/bin/bash
# A/B of environment settings on the same box.  usage: ab_env.sh "<workloads>" "VAR=a" "VAR=b" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
WL=$1; shift
for W in $WL; do
for setting in "$@"; do
  echo "== $W $setting"
  env $setting python scripts/tile_stats.py $W 2>&1 | grep -E "kernel times" | tail -1 | sed -E 's/.*(k_blend=[0-9.]+).*(total=[0-9.]+)/   \1 \2/'
  env $setting python bench.py --steps 300 --warmup 20 --no-cpu-baseline --workload $W 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('   fps', round(j['value'],1), 'single', round(j['config']['single_stream_fps'],1))"
done
done
