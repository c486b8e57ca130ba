#!/bin/bash
# quick GPU check: render+preprocess+sort parity tests, then per-kernel times and a short bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 -p no:cacheprovider 2>&1 | tail -15
python scripts/tile_stats.py ${WORKLOAD:-c2} 2>&1 | tail -6
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload ${WORKLOAD:-c2} 2>&1 | tail -1 > gpurun_out/quick_bench.json
python - <<'PY'
import json
j=json.loads(open('gpurun_out/quick_bench.json').read())
print('fps', round(j['value'],1), {k:round(v['ms']*1000,1) for k,v in j['stages'].items()})
PY
