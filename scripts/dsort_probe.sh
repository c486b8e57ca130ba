#!/bin/bash
# rocprofv3 kernel stats of scripts/dsort_probe.py for each depth-sort path and key distribution -> gpurun_out/dsort_probe/
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=gpurun_out/dsort_probe; mkdir -p $OUT; rm -f $OUT/summary.txt
for MODE in ${MODES:-adaptive twolevel}; do for DIST in ${DISTS:-uniform peaked}; do for N in ${NS:-700000 5000000}; do
  D=$OUT/${MODE}_${DIST}_$N; rm -rf $D
  WS_DEPTH_SORT=$MODE timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- python scripts/dsort_probe.py $N $DIST > $D.log 2>&1
  echo "== $MODE $DIST $N  $(tail -1 $D.log)" >> $OUT/summary.txt
  python - $D/p_kernel_stats.csv >> $OUT/summary.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    nm = r["Name"].split("(")[0].split("::")[-1]
    if "sort" in nm or "minmax" in nm or "copy" in nm:
        print("   %-34s calls %4s avg %8.1f us" % (nm[:34], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  find $D -name "*trace*" -delete
done; done; done
cat $OUT/summary.txt
