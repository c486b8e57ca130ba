#!/bin/bash
# A/B of environment settings on one box: frames/s (4 in flight / single stream) per workload and setting.
#   SETTINGS="default|WS_DEPTH_SORT=classic"  WORKLOADS="hd1m c3"  TAG=ab1
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-ab1}
OUT=gpurun_out/r02_$TAG
mkdir -p $OUT
IFS='|' read -ra SETS <<< "${SETTINGS:-default}"
for W in ${WORKLOADS:-hd1m}; do
  for S in "${SETS[@]}"; do
    name=$(echo "$S" | tr ' =/' '___')
    if [[ "$S" == "default" ]]; then E=""; else E="$S"; fi
    env $E timeout 600 python bench.py --steps ${STEPS:-600} --warmup 50 --workload $W --no-cpu-baseline --no-dist > $OUT/ab_${W}_${name}.json 2> $OUT/ab_${W}_${name}.err
    python - "$OUT/ab_${W}_${name}.json" "$W" "$S" <<'PY' >> $OUT/ab_summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).readline())
    k = d["kernels"]
    depth = sum(v["ms_per_frame"] for n, v in k.items() if n.startswith("depth:"))
    print(f"{sys.argv[2]:6s} {sys.argv[3]:40s} fps {d['value']:8.0f} single {d['config']['single_stream_fps']:8.0f}  stages "
          + " ".join(f"{n[:4]} {v['ms']*1e3:6.1f}" for n, v in d["stages"].items()))
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
  done
done
cat $OUT/ab_summary.txt
