#!/bin/bash
# rocprofv3 kernel stats of bench.py (one frame in flight and four) under an environment setting.
#   prof_variant.sh <tag> <workload> [ENV=VALUE ...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=$1; W=$2; shift 2
OUT=gpurun_out/r03_prof_$TAG; mkdir -p $OUT
for S in 1 4; do
  rm -rf $OUT/p$S
  env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p$S -o prof -- python bench.py --steps 200 --warmup 20 --streams $S --workload $W --no-cpu-baseline --no-dist > $OUT/p$S.log 2>&1
  cp $OUT/p$S/prof_kernel_stats.csv $OUT/${W}_${TAG}_streams${S}_kernel_stats.csv
  find $OUT/p$S -name "*kernel_trace*" -size +4M -delete
done
python - $OUT/${W}_${TAG}_streams1_kernel_stats.csv $OUT/${W}_${TAG}_streams4_kernel_stats.csv <<'PY'
import csv,sys
for f in sys.argv[1:]:
    print(f)
    for r in list(csv.DictReader(open(f)))[:12]:
        print("  ", r["Name"][:64], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
