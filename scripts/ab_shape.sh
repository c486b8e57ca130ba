#!/bin/bash
# A/B of the binning-tile shapes on the same box: per-kernel times + fps.  usage: ab_shape.sh <workload>...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for W in "$@"; do
for shape in 2x2 4x2 4x4; do
  echo "== $W $shape"
  WS_TILE_SHAPE=$shape python scripts/tile_stats.py $W 2>&1 | grep -E "kernel times|D=" | tail -2
  WS_TILE_SHAPE=$shape python bench.py --steps 300 --warmup 20 --no-cpu-baseline --workload $W 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('fps', round(j['value'],1), 'single', round(j['config']['single_stream_fps'],1))"
done
done
