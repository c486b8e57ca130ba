#!/bin/bash
# A/B of library BUILDS (web-splat_amd/lib_<x>/, made with `make EXTRA=-D... OBJDIR=... LIB=...`) and environment settings
# on one box.   CASES="name|ENV=..;ENV=..|libdir" separated by spaces, e.g. CASES="base||lib s256||lib_s256 pad60|WS_BLEND_LDS_PAD_KB=60|lib"
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-abl}
OUT=gpurun_out/r02_$TAG
mkdir -p $OUT
for W in ${WORKLOADS:-hd1m}; do
  for C in $CASES; do
    IFS='|' read -r name envs libdir <<< "$C"
    E=$(echo "$envs" | tr ';' ' ')
    env $E WEBSPLAT_LIB=$PWD/web-splat_amd/${libdir:-lib}/libwebsplat_hip.so timeout 600 python bench.py --steps ${STEPS:-600} --warmup 50 --workload $W --no-cpu-baseline --no-dist > $OUT/abl_${W}_${name}.json 2> $OUT/abl_${W}_${name}.err
    python - "$OUT/abl_${W}_${name}.json" "$W" "$name" <<'PY' >> $OUT/abl_summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).readline())
    print(f"{sys.argv[2]:6s} {sys.argv[3]:24s} fps {d['value']:8.0f} single {d['config']['single_stream_fps']:8.0f}  stages "
          + " ".join(f"{n[:4]} {v['ms']*1e3:6.1f}" for n, v in d["stages"].items()))
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
  done
done
cat $OUT/abl_summary.txt
