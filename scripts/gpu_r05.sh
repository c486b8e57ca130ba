#!/bin/bash
# Round-5 GPU batch (run through gpurun).  Output lands in gpurun_out/r05_<TAG>/; what should be judged is copied to profiles/r05/.
#   WHAT=host,selflaunch,blendwait,tests,ab,prof,...   TAG=a
#   hostvar     bench.py (hd1m, 2000 frames) under host-side variants: who burns the host cores of a rank (config.host_threads)
#   selflaunch  `python bench.py --gpus 2 --single-device --dist-backend gloo` from a plain shell (no launcher in the command)
#   blendwait   scripts/blend_wait_breakdown.py on hd1m and c3 (time-stamped build of k_blend)
#   tests       pytest -m gpu (PYTEST_ARGS narrows it)
#   abtest      bench.py per variant of VARIANTS ("name=ENV=VAL,ENV=VAL ...") and workload of WORKLOADS: frames/s in flight / alone
#   prof        rocprofv3 --kernel-trace --stats of bench.py --streams 1 per workload of WORKLOADS
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-a}
OUT=gpurun_out/r05_$TAG
mkdir -p $OUT
WHAT=${WHAT:-host,selflaunch,blendwait}
WORKLOADS=${WORKLOADS:-hd1m c3}
STEPS=${STEPS:-1000}
rm -f $OUT/summary.txt
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6 > $OUT/device.txt
nproc >> $OUT/device.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/device.txt 2>/dev/null
for c in /sys/class/drm/card*/device/numa_node; do echo "$c $(cat $c)"; done >> $OUT/device.txt 2>/dev/null
python - >> $OUT/device.txt 2>&1 <<'PY'
import sys; sys.path.insert(0, ".")
import bench
print("gpu_numa_nodes", bench.gpu_numa_nodes(), "visible", bench._visible_device_ordinals())
print("numa_share(0,1)", bench.numa_share(0, 1, sorted(__import__("os").sched_getaffinity(0)))[1])
print("numa_share(0,8)", bench.numa_share(0, 8, sorted(__import__("os").sched_getaffinity(0)), gpu_nodes=None)[1])
PY

line() {  # $1 = json file -> one summary line
python - "$1" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = j["config"]
    thr = c.get("host_threads") or []
    print("fps", round(j["value"], 1), "single", round(c.get("single_stream_fps", 0), 1), "enq_ms", round(c["host_enqueue_ms_per_frame"], 4),
          "cores_busy", round(c["host_cores_busy_per_rank"], 3), "wait", c.get("host_wait"), "submit", c.get("frame_submission"),
          "threads", [(t["thread"], t["cores"]) for t in thr[:6]])
except Exception as e:
    print("unreadable:", e)
PY
}

if [[ $WHAT == *hostvar* ]]; then
  i=0
  for v in "spin=--host-wait spin" "block=--host-wait block" "block_graph=--host-wait block|WS_GRAPH=1" "spin_graph=--host-wait spin|WS_GRAPH=1" \
           "block_awt0=--host-wait block|ROC_ACTIVE_WAIT_TIMEOUT=0" "block_nodirect=--host-wait block|AMD_DIRECT_DISPATCH=0" \
           "block_graph_s1=--host-wait block --streams 1|WS_GRAPH=1" "block_s1=--host-wait block --streams 1" ${HOST_EXTRA:-}; do
    N=${v%%=*}; rest=${v#*=}; args=${rest%%|*}; envs=""; [[ $rest == *"|"* ]] && envs=${rest#*|}
    env ${envs//,/ } timeout 300 python bench.py --steps ${HOST_STEPS:-2000} --warmup 50 --no-secondary --no-cpu-baseline $args \
        > $OUT/host_$N.json 2> $OUT/host_$N.err
    echo "host $N exit=$? : $(line $OUT/host_$N.json)" >> $OUT/summary.txt
  done
fi
if [[ $WHAT == *graphstab* ]]; then
  # the default invocation (with its c3 block and CPU baseline) five times under WS_GRAPH=1: does the ROCm fault show?
  for k in 1 2 3 4 5; do
    WS_GRAPH=1 timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/graphstab_$k.json 2> $OUT/graphstab_$k.err
    echo "graphstab $k exit=$? : $(line $OUT/graphstab_$k.json)" >> $OUT/summary.txt
  done
fi
if [[ $WHAT == *selflaunch* ]]; then
  timeout 600 python bench.py --gpus 2 --single-device --dist-backend gloo --workload c2 --steps 30 --warmup 5 --no-cpu-baseline \
      > $OUT/selflaunch.json 2> $OUT/selflaunch.err
  echo "selflaunch exit=$? lines=$(grep -c '^{' $OUT/selflaunch.json) n_gpus=$(python -c "import json;print(json.loads(open('$OUT/selflaunch.json').read().strip().splitlines()[-1])['n_gpus'])" 2>&1)" >> $OUT/summary.txt
fi
if [[ $WHAT == *blendwait* ]]; then
  for W in ${BW_WORKLOADS:-hd1m c3}; do
   for O in ${BW_ORDER:-1}; do
    WS_BLEND_ORDER=$O timeout 600 python scripts/blend_wait_breakdown.py $W 0 $OUT/blend_wait_breakdown_${W}_order$O.json > $OUT/blendwait_${W}_order$O.log 2>&1
    echo "blendwait $W order=$O exit=$?" >> $OUT/summary.txt
    python - $OUT/blend_wait_breakdown_${W}_order$O.json >> $OUT/summary.txt 2>&1 <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print("  kernel us prod/timing", round(j["kernel_us_production_build"], 1), round(j["kernel_us_timing_build"], 1), "span", round(j["span_us_first_start_to_last_end"], 1),
      "slot time", round(j["slot_time_us (sum of tile durations / slots)"], 1), "mean resident", round(j["mean_tiles_resident"], 1), "MHz", round(j["shader_clock_mhz"]),
      "identical", j["image_identical_to_production_build"])
print("  share us:", {k: round(v, 1) for k, v in j["kernel_share_us"].items()})
for t in ("median_tile", "p99_tile"):
    r = j[t]
    print(" ", t, "dur", round(r["tile_duration_us"], 1), "list", r.get("list_len"), "batches", r["batches"], "walked", round(r["records_walked_mean_wave"]), r["records_walked_max_wave"],
          {k: round(v, 2) for k, v in r["mean_wave_us"].items()}, "other", round(r["other_us"], 2))
print("  walk:", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in j["walk"].items() if k != "note"})
print("  tail/idle", round(j["tail_and_idle_us (span - slot time)"], 1), "dur pct", {k: round(v, 1) for k, v in j["tile_duration_us_pct"].items()})
for k, v in j["mean_tile_us_by_wave_role"].items():
    print("  ", k, {a: round(b, 2) for a, b in v.items()})
PY
   done
  done
fi
if [[ $WHAT == *hostprobe* ]]; then
  bash scripts/host_thread_probe.sh $OUT/hostprobe ${HOSTPROBE_ARGS:-} > $OUT/hostprobe.txt 2>&1
  echo "hostprobe exit=$?" >> $OUT/summary.txt; head -60 $OUT/hostprobe.txt >> $OUT/summary.txt
fi
if [[ $WHAT == *tests* ]]; then
  timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --maxfail ${MAXFAIL:-5} ${PYTEST_ARGS:-} > $OUT/tests_gpu_full.log 2>&1; tail -60 $OUT/tests_gpu_full.log > $OUT/tests_gpu.log
  echo "tests exit=$?" >> $OUT/summary.txt; tail -3 $OUT/tests_gpu.log >> $OUT/summary.txt
  cp gpurun_out/parity_fullsize.json $OUT/ 2>/dev/null
fi
if [[ $WHAT == *exptests* ]]; then
  # the measured-and-lost variants live in the experimental build only: their tests against lib_exp
  WEBSPLAT_LIB=$PWD/web-splat_amd/lib_exp/libwebsplat_hip.so timeout 1800 python -m pytest tests/test_gpu_render.py tests/test_gpu_sort.py \
      tests/test_gpu_fullsize.py -m gpu -q --timeout 900 -p no:cacheprovider --maxfail 10 ${EXP_PYTEST_ARGS:-} > $OUT/tests_gpu_experimental_full.log 2>&1
  echo "exptests exit=$?" >> $OUT/summary.txt; tail -3 $OUT/tests_gpu_experimental_full.log >> $OUT/summary.txt
fi
if [[ $WHAT == *cutoff* ]]; then
  timeout 900 python scripts/cutoff_study.py $OUT/cutoff_study_c4.json > $OUT/cutoff.log 2>&1
  echo "cutoff exit=$?" >> $OUT/summary.txt
  python - $OUT/cutoff_study_c4.json >> $OUT/summary.txt 2>&1 <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print("  consumed frac", round(j["consumed_frac_of_D"], 3), "saturated tiles", round(j["saturated_tiles_frac"], 3), "D/view", round(j["entries_per_view"]))
for r in j["table"]:
    print("  delta", r["delta_views"], "margin", r["margin"], "removable", round(r["removable_frac_of_D"], 3), "redrawn tiles", round(r["redrawn_tiles_frac"], 4),
          "entries in redrawn", round(r["entries_in_redrawn_tiles_frac_of_D"], 3))
PY
fi
if [[ $WHAT == *abtest* ]]; then
  for W in $WORKLOADS; do
    for v in ${VARIANTS:-base=}; do
      N=${v%%=*}; envs=${v#*=}
      env ${envs//,/ } timeout 600 python bench.py --workload $W --steps $STEPS --warmup 50 --no-secondary --no-cpu-baseline ${AB_ARGS:-} \
          > $OUT/ab_${N}_$W.json 2> $OUT/ab_${N}_$W.err
      echo "ab $N $W exit=$? : $(line $OUT/ab_${N}_$W.json)" >> $OUT/summary.txt
      python - $OUT/ab_${N}_$W.json >> $OUT/summary.txt 2>&1 <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("    kernels us:", {k: round(v["ms_per_frame"] * 1e3, 1) for k, v in j["kernels"].items()}, "stages", {k: round(v["ms"] * 1e3, 1) for k, v in j["stages"].items()})
except Exception as e:
    print("    unreadable:", e)
PY
    done
  done
fi
if [[ $WHAT == *prof* ]]; then
  for W in $WORKLOADS; do
    for v in ${PROF_VARIANTS:-final=}; do
      N=${v%%=*}; envs=${v#*=}
      rm -rf /tmp/prof_$N_$W
      ( cd /tmp && env ${envs//,/ } timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_${N}_$W -o p -- python $OLDPWD/bench.py --workload $W --streams 1 \
          --steps 300 --warmup 20 --no-secondary --no-cpu-baseline > $OLDPWD/$OUT/prof_${N}_$W.json 2> $OLDPWD/$OUT/prof_${N}_$W.err )
      echo "prof $N $W exit=$?" >> $OUT/summary.txt
      f=$(find /tmp/prof_${N}_$W -name "*kernel_stats.csv" | head -1)
      [[ -n $f ]] && cp $f $OUT/${W}_${N}_kernel_stats.csv && head -14 $f | cut -d, -f1-6 >> $OUT/summary.txt
    done
  done
fi
cat $OUT/summary.txt
