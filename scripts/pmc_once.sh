#!/bin/bash
# usage: pmc_once.sh <tag> "<COUNTERS...>" [bench args]  -> gpurun_out/pmc_<tag>/  (counters in their own run: no --stats/sys-trace)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=$1; CTRS=$2; shift; shift
rm -rf gpurun_out/pmc_$TAG
timeout 600 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d gpurun_out/pmc_$TAG -o pmc -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/pmc_$TAG.log 2>&1
python scripts/pmc_summary.py gpurun_out/pmc_$TAG/pmc_counter_collection.csv
