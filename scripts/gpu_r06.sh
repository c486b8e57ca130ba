#!/bin/bash
# Round-6 GPU batch (run through gpurun).  Output lands in gpurun_out/r06_<TAG>/; what should be judged is copied to profiles/r06/.
#   WHAT=ubenchq,inflight,tests,abtest,prof,...   TAG=a
#   ubenchq    scripts/ubench/queue_concurrency under GPU_MAX_HW_QUEUES = 4, 8, 16: how many streams run side by side
#   inflight   rocprofv3 kernel trace of bench.py --streams S for S of STREAMS -> scripts/inflight_overlap.py (queue / label overlap)
#   tests      pytest -m gpu (PYTEST_ARGS narrows it)
#   abtest     bench.py per variant of VARIANTS ("name=ENV=VAL,ENV=VAL ..."; WEBSPLAT_LIB=... selects a library build) and workload
#   prof       rocprofv3 --kernel-trace --stats of bench.py --streams 1 per workload of WORKLOADS (and per variant of PROF_VARIANTS)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-a}
OUT=gpurun_out/r06_$TAG
mkdir -p $OUT
WHAT=${WHAT:-ubenchq,inflight}
WORKLOADS=${WORKLOADS:-hd1m c3}
STEPS=${STEPS:-1000}
rm -f $OUT/summary.txt
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6 > $OUT/device.txt
nproc >> $OUT/device.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/device.txt 2>/dev/null

line() {  # $1 = json file -> one summary line
python - "$1" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = j["config"]
    st = j.get("stages") or {}
    print("fps", round(j["value"], 1), "single", round(c.get("single_stream_fps", 0), 1), "enq_ms", round(c["host_enqueue_ms_per_frame"], 4),
          "us:", {k[:5]: round(1e3 * v["ms"], 1) for k, v in st.items()})
except Exception as e:
    print("unreadable:", e)
PY
}

if [[ $WHAT == *ubenchq* ]]; then
  for Q in 4 8 16; do
    QC_TIMELINE=1 GPU_MAX_HW_QUEUES=$Q timeout 300 scripts/ubench/queue_concurrency ${UBENCH_STREAMS:-12} > $OUT/queue_concurrency_q$Q.jsonl 2> $OUT/queue_concurrency_q$Q.err
    echo "ubenchq q=$Q exit=$?" >> $OUT/summary.txt
  done
fi

if [[ $WHAT == *inflight* ]]; then
  for W in ${INFLIGHT_WORKLOADS:-hd1m}; do
  for S in ${STREAMS:-4 5 6 8}; do
    D=$OUT/trace_${W}_s$S
    rm -rf $D
    timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python bench.py --workload $W --streams $S --steps ${TRACE_STEPS:-300} --warmup 30 \
        --no-secondary --no-cpu-baseline > $OUT/trace_${W}_s$S.log 2>&1
    T=$(find $D -name "*kernel_trace.csv" | head -1)
    python scripts/inflight_overlap.py "$T" 40 --json $OUT/inflight_overlap_${W}_s$S.json > /dev/null 2> $OUT/inflight_overlap_${W}_s$S.err
    echo "inflight $W s=$S exit=$? : $(python -c "
import json; j=json.load(open('$OUT/inflight_overlap_${W}_s$S.json'))
print('us/frame', round(j['us_per_frame'],1), 'conc', round(j['mean_concurrency'],2), 'idle', round(j['idle_us_per_frame'],1), 'never', j['queue_pairs_that_never_overlap'], j['wall_share_by_what_runs'])" 2>&1)" >> $OUT/summary.txt
    rm -rf $D   # (the raw trace is tens of MB; the reduction is what is kept)
    # the same without the tracer: the rate the shape belongs to
    timeout 300 python bench.py --workload $W --streams $S --steps $STEPS --warmup 50 --no-secondary --no-cpu-baseline > $OUT/bench_${W}_s$S.json 2> $OUT/bench_${W}_s$S.err
    echo "bench $W s=$S exit=$? : $(line $OUT/bench_${W}_s$S.json)" >> $OUT/summary.txt
  done
  done
fi

if [[ $WHAT == *tests* ]]; then
  timeout ${TEST_TIMEOUT:-2400} python -m pytest tests -m gpu -x -q --timeout 900 -p no:cacheprovider ${PYTEST_ARGS:-} > $OUT/tests_gpu.log 2>&1
  echo "tests exit=$? : $(tail -1 $OUT/tests_gpu.log)" >> $OUT/summary.txt
fi

if [[ $WHAT == *abtest* ]]; then
  for rep in $(seq 1 ${AB_REPS:-2}); do
  for W in $WORKLOADS; do
  for v in ${VARIANTS:-base=}; do
    N=${v%%=*}; envs=${v#*=}
    env ${envs//,/ } timeout 300 python bench.py --workload $W --steps $STEPS --warmup 50 --no-secondary --no-cpu-baseline ${AB_ARGS:-} \
        > $OUT/ab_${N}_${W}_$rep.json 2> $OUT/ab_${N}_${W}_$rep.err
    echo "ab $N $W rep=$rep exit=$? : $(line $OUT/ab_${N}_${W}_$rep.json)" >> $OUT/summary.txt
  done
  done
  done
fi

if [[ $WHAT == *prof* ]]; then
  for W in $WORKLOADS; do
  for v in ${PROF_VARIANTS:-base=}; do
    N=${v%%=*}; envs=${v#*=}
    D=$OUT/prof_${N}_$W
    rm -rf $D
    env ${envs//,/ } timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o prof -- python bench.py --workload $W --streams 1 --steps 50 --warmup 10 \
        --no-secondary --no-cpu-baseline > $D.log 2>&1
    S=$(find $D -name "*kernel_stats.csv" | head -1)
    [[ -n "$S" ]] && cp "$S" $OUT/${W}_${N}_kernel_stats.csv
    T=$(find $D -name "*kernel_trace.csv" | head -1)
    [[ -n "$T" ]] && python scripts/frame_timeline.py "$T" > $OUT/${W}_${N}_frame_timeline.txt 2>&1
    rm -rf $D
    echo "prof $N $W : $(python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/${W}_${N}_kernel_stats.csv")))
out = []
for r in rows[:9]:
    out.append(r["Name"].split("(")[0].split("<")[0][-18:] + "=" + str(round(float(r["AverageNs"]) / 1e3, 1)) + "x" + r["Calls"])
print(" ".join(out))
PY
)" >> $OUT/summary.txt
  done
  done
fi
cat $OUT/summary.txt
