#!/bin/bash
# A/B two builds of the library on the same box: per-kernel times + fps.  usage: ab_lib.sh <libdir> [workload]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
W=${2:-c2}
for lib in web-splat_amd/lib/libwebsplat_hip.so web-splat_amd/$1/libwebsplat_hip.so; do
  echo "== $lib"
  WEBSPLAT_LIB=$PWD/$lib python scripts/tile_stats.py $W 2>&1 | grep "kernel times" | tail -1
  WEBSPLAT_LIB=$PWD/$lib python bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload $W 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('fps', round(j['value'],1), {k:round(v['ms']*1000,1) for k,v in j['stages'].items()})"
done
