#!/bin/bash
# Round-4 GPU batch (run through gpurun).  Everything lands in gpurun_out/r04_<TAG>/; copy what should be judged into
# profiles/r04/.
#   VARIANTS="name=libdir[,ENV=VALUE...] ..."   e.g. "base=lib_base rect=lib ellipse=lib,WS_FOOTPRINT=ellipse dma=lib,WS_BLEND_DMA=1"
#   WHAT=tests,equal,ab,prof  WORKLOADS="hd1m c3"  TAG=a  STEPS=600
#   tests  pytest -m gpu per variant of TEST_VARIANTS (PYTEST_ARGS_<variant> or PYTEST_ARGS narrow it)
#          [was:] pytest -m gpu per variant of TEST_VARIANTS (default: the first of VARIANTS named in it; PYTEST_ARGS narrows it)
#   equal  scripts/dump_images.py per variant, images compared bit for bit against the FIRST variant
#   ab     per variant and workload: per-kernel event times of one frame (scripts/tile_stats.py) + bench.py frames/s
#   prof   rocprofv3 --kernel-trace --stats of bench.py (one frame in flight) for the variants of PROF_VARIANTS
#   ubench scripts/ubench/grid_barrier (device-wide barrier against the kernel boundary it replaces)
#   measure scripts/measure_procedure.py (the reference's bin/measure.rs procedure on the c2 stand-in)
#   host   the host-contention test of bench.py + the default bench line with its c3 block
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-a}
OUT=gpurun_out/r04_$TAG
mkdir -p $OUT
STEPS=${STEPS:-600}
WORKLOADS=${WORKLOADS:-hd1m c3}
VARIANTS=${VARIANTS:-"base=lib_base new=lib"}
WHAT=${WHAT:-tests,equal,ab,prof}
rm -f $OUT/summary.txt $OUT/ab.txt
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6 > $OUT/device.txt
# variant spec -> "env ..." prefix
venv() {  # $1 = spec "lib[,K=V...]"
  local spec=$1 lib=${1%%,*} rest=""
  [[ $spec == *,* ]] && rest=${spec#*,}
  echo "WEBSPLAT_LIB=$PWD/web-splat_amd/$lib/libwebsplat_hip.so ${rest//,/ }"
}
vspec() { for v in $VARIANTS; do [[ ${v%%=*} == ${1:-} ]] && echo ${v#*=}; done; true; }
if [[ $WHAT == *ubench* ]]; then
  timeout 300 scripts/ubench/grid_barrier > $OUT/grid_barrier.jsonl 2> $OUT/grid_barrier.err; echo "ubench grid_barrier exit=$?" >> $OUT/summary.txt
  head -3 $OUT/grid_barrier.jsonl >> $OUT/summary.txt
fi
if [[ $WHAT == *measure* ]]; then
  timeout 900 python scripts/measure_procedure.py $OUT/measure_rs_procedure.json > $OUT/measure.log 2>&1; echo "measure exit=$?" >> $OUT/summary.txt
  grep -E "average_fps_(best|median)" $OUT/measure.log >> $OUT/summary.txt
fi
if [[ $WHAT == *host* ]]; then
  timeout 900 python -m pytest tests/test_gpu_bench.py -m gpu -q --timeout 900 -p no:cacheprovider -k "sliver or secondary or contract or capacity" 2>&1 | tail -15 > $OUT/tests_host.log
  echo "host tests exit=${PIPESTATUS[0]}" >> $OUT/summary.txt; tail -3 $OUT/tests_host.log >> $OUT/summary.txt
  cp gpurun_out/host_contention.json $OUT/ 2>/dev/null; cat $OUT/host_contention.json >> $OUT/summary.txt 2>/dev/null
  ( time timeout 600 python bench.py > $OUT/default_bench.json 2> $OUT/default_bench.err ) 2> $OUT/default_bench.time; echo "default bench exit=$?" >> $OUT/summary.txt
  grep real $OUT/default_bench.time >> $OUT/summary.txt
  python - $OUT/default_bench.json >> $OUT/summary.txt <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=j["config"]; print("default: fps", round(j["value"],1), "single", round(c["single_stream_fps"],1), "enq_ms", round(c["host_enqueue_ms_per_frame"],4), "host_bound", c["host_bound"], c["host_affinity"])
    s=j["secondary"]["c3"]; print("c3: fps", round(s["value"],1), "single", round(s["single_stream_fps"],1), "roofline", s["roofline"]["kernel"], s["roofline"]["frac"], {k:(round(v["ms_per_frame"],4), round(v["frac"],3)) for k,v in s["kernels"].items()})
except Exception as e:
    print("default bench unreadable:", e)
PY
fi
if [[ $WHAT == *tests* ]]; then
  for N in ${TEST_VARIANTS:-$(echo $VARIANTS | awk '{print $NF}' | cut -d= -f1)}; do
    av=PYTEST_ARGS_$N   # per-variant pytest arguments (e.g. PYTEST_ARGS_dma="-k image"), else PYTEST_ARGS
    env $(venv $(vspec $N)) timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --maxfail ${MAXFAIL:-5} ${!av:-${PYTEST_ARGS:-}} 2>&1 | tail -40 > $OUT/tests_gpu_$N.log
    echo "tests $N exit=${PIPESTATUS[0]}" >> $OUT/summary.txt; tail -3 $OUT/tests_gpu_$N.log >> $OUT/summary.txt
  done
  cp gpurun_out/parity_fullsize.json $OUT/ 2>/dev/null
fi
if [[ $WHAT == *equal* ]]; then
  first=""
  for v in $VARIANTS; do
    N=${v%%=*}
    for W in ${EQUAL_WORKLOADS:-$WORKLOADS}; do
      env $(venv ${v#*=}) timeout 600 python scripts/dump_images.py $W /tmp/img_$N 0 3 > $OUT/dump_${N}_$W.log 2>&1 || echo "dump $N $W FAILED" >> $OUT/summary.txt
    done
    if [[ -z $first ]]; then first=$N; else
      python scripts/dump_images.py --compare /tmp/img_$first /tmp/img_$N > $OUT/equal_${first}_vs_$N.txt 2>&1; echo "equal $first vs $N exit=$?" >> $OUT/summary.txt
    fi
  done
fi
if [[ $WHAT == *ab* ]]; then
  for W in $WORKLOADS; do
    for v in $VARIANTS; do
      N=${v%%=*}
      echo "== $W $N (${v#*=})" >> $OUT/ab.txt
      [[ -z ${SKIP_TILE_STATS:-} ]] && env $(venv ${v#*=}) timeout 600 python scripts/tile_stats.py $W 2>&1 | grep -E "^view 0|^  consumed|kernel times" | head -3 >> $OUT/ab.txt
      env $(venv ${v#*=}) timeout 600 python bench.py --steps $STEPS --warmup 30 --no-cpu-baseline --no-secondary --workload $W 2> $OUT/bench_${N}_$W.err | tail -1 > $OUT/bench_${N}_$W.json
      python - $OUT/bench_${N}_$W.json >> $OUT/ab.txt <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read())
    k=j["kernels"]
    depth = sum(v["ms_per_frame"] for l, v in k.items() if l.startswith("depth:"))
    print("fps", round(j["value"],1), "single", round(j["config"]["single_stream_fps"],1), "V", int(j["config"]["avg_visible"]), "D", int(j["config"]["avg_tile_entries"]),
          "blend_us", round((k["k_blend"]["avg_launch_ms"] or 0)*1e3,1), "depth_sort_us(events)", round(depth*1e3,1),
          "enq_ms", round(j["config"]["host_enqueue_ms_per_frame"],4))
except Exception as e:
    print("bench failed:", e)
PY
    done
  done
  cat $OUT/ab.txt >> $OUT/summary.txt
fi
if [[ $WHAT == *pmcvalu* ]]; then   # VALU issue / wait accounting per kernel (one PMC pass; no trace domains beside --kernel-trace)
  for W in ${PMC_WORKLOADS:-c3}; do
    rm -rf $OUT/pmc_valu_$W
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_valu_$W -o pmc -- python bench.py --steps 6 --warmup 2 --streams 1 --no-cpu-baseline --no-secondary --no-dist --workload $W > $OUT/pmc_valu_$W.log 2>&1
    python scripts/pmc_valu.py $OUT/pmc_valu_$W/pmc_counter_collection.csv $OUT/valu_$W.json > $OUT/valu_$W.txt 2>&1; echo "pmc valu $W exit=$?" >> $OUT/summary.txt
    rm -rf $OUT/pmc_lds_$W
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_lds_$W -o pmc -- python bench.py --steps 6 --warmup 2 --streams 1 --no-cpu-baseline --no-secondary --no-dist --workload $W > $OUT/pmc_lds_$W.log 2>&1
    python scripts/pmc_summary.py $OUT/pmc_lds_$W/pmc_counter_collection.csv > $OUT/lds_$W.txt 2>&1; echo "pmc lds $W exit=$?" >> $OUT/summary.txt
    find $OUT/pmc_valu_$W $OUT/pmc_lds_$W -size +2M -delete 2>/dev/null
  done
fi
if [[ $WHAT == *prof* ]]; then
  for N in ${PROF_VARIANTS:-$(echo $VARIANTS | awk '{print $NF}' | cut -d= -f1)}; do
    for W in ${PROF_WORKLOADS:-$WORKLOADS}; do
      rm -rf $OUT/prof_${N}_$W
      env $(venv $(vspec $N)) timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${N}_$W -o prof -- python bench.py --steps 60 --warmup 10 --streams 1 --workload $W --no-cpu-baseline --no-secondary --no-dist > $OUT/prof_${N}_$W.log 2>&1; echo "prof $W ($N) exit=$?" >> $OUT/summary.txt
      python scripts/frame_timeline.py $OUT/prof_${N}_$W/prof_kernel_trace.csv > $OUT/${W}_${N}_frame_timeline.txt 2>&1
      cp $OUT/prof_${N}_$W/prof_kernel_stats.csv $OUT/${W}_${N}_kernel_stats.csv 2>/dev/null
      find $OUT/prof_${N}_$W -name "*kernel_trace*" -size +4M -delete
    done
  done
fi
cat $OUT/summary.txt
