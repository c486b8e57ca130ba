#!/bin/bash
# A/B builds of the library on the same box: per-kernel times + fps.  usage: ab_lib.sh <workload> <libdir>...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
W=${1:-c2}; shift
for d in lib "$@"; do
  lib=web-splat_amd/$d/libwebsplat_hip.so
  echo "== $lib"
  WEBSPLAT_LIB=$PWD/$lib python scripts/tile_stats.py $W 2>&1 | grep "kernel times" | tail -1
  WEBSPLAT_LIB=$PWD/$lib python bench.py --steps 300 --warmup 20 --no-cpu-baseline --workload $W 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('fps', round(j['value'],1), 'single', round(j['config']['single_stream_fps'],1))"
done
