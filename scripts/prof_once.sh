#!/bin/bash
# usage: prof_once.sh <tag> [bench args...]   (env vars pass through) -> gpurun_out/prof_<tag>/
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=$1; shift
rm -rf gpurun_out/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o prof -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/prof_$TAG.log 2>&1
python scripts/frame_timeline.py gpurun_out/prof_$TAG/prof_kernel_trace.csv
