cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_batchprof; mkdir -p $OUT
for G in 1 4; do
  rm -rf $OUT/prof_$G
  WS_BATCH_K1=$G timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$G -o prof -- python bench.py --steps 200 --warmup 20 --streams 4 --workload hd1m --no-cpu-baseline --no-dist > $OUT/prof_$G.log 2>&1
  cp $OUT/prof_$G/prof_kernel_stats.csv $OUT/hd1m_batch${G}_inflight_kernel_stats.csv
  tail -1 $OUT/prof_$G.log | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('G=$G fps', round(j['value'],1))"
  find $OUT/prof_$G -name "*kernel_trace*" -size +4M -delete
done
python - <<'PY'
import csv
for G in (1,4):
    rows=list(csv.DictReader(open(f"gpurun_out/r03_batchprof/hd1m_batch{G}_inflight_kernel_stats.csv")))
    print("G",G)
    for r in rows[:8]:
        print("  ", r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
