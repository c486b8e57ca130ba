#!/bin/bash
# Round-3 GPU batch (run through gpurun).  Everything lands in gpurun_out/r03_<TAG>/; copy what should be judged into
# profiles/r03/.   WHAT=tests,equal,ab,prof  LIBS="lib_base lib"  WORKLOADS="hd1m c3"  TAG=a  STEPS=600
#   tests  pytest -m gpu on the default library (PYTEST_ARGS narrows it; TEST_LIBS="lib_x ..." repeats it per build)
#   equal  scripts/dump_images.py per library, images compared bit for bit against the FIRST library of LIBS
#   ab     per library and workload: per-kernel event times of one frame (scripts/tile_stats.py) + bench.py frames/s
#   prof   rocprofv3 --kernel-trace --stats of bench.py (one frame in flight) on the LAST library of LIBS
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-a}
OUT=gpurun_out/r03_$TAG
mkdir -p $OUT
STEPS=${STEPS:-600}
WORKLOADS=${WORKLOADS:-hd1m c3}
LIBS=${LIBS:-lib_base lib}
WHAT=${WHAT:-tests,equal,ab,prof}
rm -f $OUT/summary.txt
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6 > $OUT/device.txt
libpath() { echo "$PWD/web-splat_amd/$1/libwebsplat_hip.so"; }
if [[ $WHAT == *tests* ]]; then
  for L in ${TEST_LIBS:-lib}; do
    WEBSPLAT_LIB=$(libpath $L) timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x ${PYTEST_ARGS:-} 2>&1 | tail -40 > $OUT/tests_gpu_$L.log
    echo "tests $L exit=${PIPESTATUS[0]}" >> $OUT/summary.txt; tail -3 $OUT/tests_gpu_$L.log >> $OUT/summary.txt
  done
  cp gpurun_out/parity_fullsize.json $OUT/ 2>/dev/null
fi
if [[ $WHAT == *equal* ]]; then
  first=""
  for L in $LIBS; do
    for W in ${EQUAL_WORKLOADS:-$WORKLOADS}; do
      WEBSPLAT_LIB=$(libpath $L) timeout 600 python scripts/dump_images.py $W /tmp/img_$L 0 3 > $OUT/dump_${L}_$W.log 2>&1 || echo "dump $L $W FAILED" >> $OUT/summary.txt
    done
    if [[ -z $first ]]; then first=$L; else
      python scripts/dump_images.py --compare /tmp/img_$first /tmp/img_$L > $OUT/equal_${first}_vs_$L.txt 2>&1; echo "equal $first vs $L exit=$?" >> $OUT/summary.txt
    fi
  done
fi
if [[ $WHAT == *ab* ]]; then
  for W in $WORKLOADS; do
    for L in $LIBS; do
      echo "== $W $L" >> $OUT/ab.txt
      WEBSPLAT_LIB=$(libpath $L) timeout 600 python scripts/tile_stats.py $W 2>&1 | grep -E "^view 0|consumed|kernel times" | head -3 >> $OUT/ab.txt
      WEBSPLAT_LIB=$(libpath $L) timeout 600 python bench.py --steps $STEPS --warmup 30 --no-cpu-baseline --workload $W 2> $OUT/bench_${L}_$W.err | tail -1 > $OUT/bench_${L}_$W.json
      python - $OUT/bench_${L}_$W.json >> $OUT/ab.txt <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read())
    k=j["kernels"]
    print("fps", round(j["value"],1), "single", round(j["config"]["single_stream_fps"],1), "V", int(j["config"]["avg_visible"]), "D", int(j["config"]["avg_tile_entries"]),
          "blend_us", round((k["k_blend"]["avg_launch_ms"] or 0)*1e3,1))
except Exception as e:
    print("bench failed:", e)
PY
    done
  done
  cat $OUT/ab.txt >> $OUT/summary.txt
fi
if [[ $WHAT == *prof* ]]; then
  L=$(echo $LIBS | awk '{print $NF}')
  for W in ${PROF_WORKLOADS:-$WORKLOADS}; do
    rm -rf $OUT/prof_$W
    WEBSPLAT_LIB=$(libpath $L) timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$W -o prof -- python bench.py --steps 60 --warmup 10 --streams 1 --workload $W --no-cpu-baseline --no-dist > $OUT/prof_$W.log 2>&1; echo "prof $W ($L) exit=$?" >> $OUT/summary.txt
    python scripts/frame_timeline.py $OUT/prof_$W/prof_kernel_trace.csv > $OUT/${W}_${L}_frame_timeline.txt 2>&1
    cp $OUT/prof_$W/prof_kernel_stats.csv $OUT/${W}_${L}_kernel_stats.csv 2>/dev/null
    find $OUT/prof_$W -name "*kernel_trace*" -size +4M -delete
  done
fi
cat $OUT/summary.txt
