#!/bin/bash
# HBM traffic per launch (PMC, separate passes, counters only) -> gpurun_out/traffic_<workload>.json (copy to profiles/)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
W=${1:-c2}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_$c -o pmc -- python bench.py --steps 6 --warmup 2 --streams 1 --no-cpu-baseline --workload $W > gpurun_out/pmc_$c.log 2>&1
done
python scripts/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE/pmc_counter_collection.csv gpurun_out/pmc_WRITE_SIZE/pmc_counter_collection.csv gpurun_out/traffic_$W.json
