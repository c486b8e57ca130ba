#!/bin/bash
# Final round-6 evidence batch: tests, smoke, then per workload PMC traffic (calibrated factors) and VALU accounting FIRST
# (bench.py embeds them in its roofline block), bench (default flags: 4 frames in flight, CPU baseline), rocprofv3 kernel
# stats + one-frame timeline.  Everything lands in gpurun_out/r06_final/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r06_${TAG:-final}
mkdir -p $OUT
rm -f $OUT/summary.txt
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6 > $OUT/device.txt; nproc >> $OUT/device.txt
if [[ ${WHAT:-tests,evidence} == *tests* ]]; then
  timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --maxfail 10 > $OUT/tests_gpu_full.log 2>&1
  echo "tests exit=$?" >> $OUT/summary.txt; tail -15 $OUT/tests_gpu_full.log > $OUT/tests_gpu.log; tail -2 $OUT/tests_gpu.log >> $OUT/summary.txt
  # the measured-and-lost variants live in the experimental build only: their tests against lib_exp
  WEBSPLAT_LIB=$PWD/web-splat_amd/lib_exp/libwebsplat_hip.so timeout 1800 python -m pytest tests -m "gpu and experimental" -q --timeout 900 \
      -p no:cacheprovider --maxfail 10 > $OUT/tests_gpu_experimental_full.log 2>&1
  echo "experimental-build tests exit=$?" >> $OUT/summary.txt; tail -15 $OUT/tests_gpu_experimental_full.log > $OUT/tests_gpu_experimental.log; tail -2 $OUT/tests_gpu_experimental.log >> $OUT/summary.txt
  cp gpurun_out/eight_rank_stand_in.json gpurun_out/host_contention.json gpurun_out/eight_rank_host_partition.json $OUT/ 2>/dev/null
  cp gpurun_out/parity_fullsize.json gpurun_out/k1_key_report.json $OUT/ 2>/dev/null
  python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit=$?" >> $OUT/summary.txt; tail -1 $OUT/smoke.log >> $OUT/summary.txt
fi
for W in ${WORKLOADS:-hd1m c3 c2}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/pmc_${W}_$c
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${W}_$c -o pmc -- python bench.py --steps 6 --warmup 2 --streams 1 --no-cpu-baseline --no-secondary --no-dist --workload $W > $OUT/pmc_${W}_$c.log 2>&1
  done
  python scripts/pmc_traffic.py $OUT/pmc_${W}_FETCH_SIZE/pmc_counter_collection.csv $OUT/pmc_${W}_WRITE_SIZE/pmc_counter_collection.csv $OUT/traffic_$W.json > $OUT/traffic_$W.log 2>&1; echo "traffic $W exit=$?" >> $OUT/summary.txt
  rm -rf $OUT/pmc_valu_$W
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_valu_$W -o pmc -- python bench.py --steps 6 --warmup 2 --streams 1 --no-cpu-baseline --no-secondary --no-dist --workload $W > $OUT/pmc_valu_$W.log 2>&1
  python scripts/pmc_valu.py $OUT/pmc_valu_$W/pmc_counter_collection.csv $OUT/valu_$W.json > /dev/null 2>&1; echo "valu $W exit=$?" >> $OUT/summary.txt
  find $OUT/pmc_${W}_FETCH_SIZE $OUT/pmc_${W}_WRITE_SIZE $OUT/pmc_valu_$W -size +2M -delete 2>/dev/null
  cp $OUT/traffic_$W.json $OUT/valu_$W.json profiles/ 2>/dev/null   # (on the box: read by the bench run below)
  timeout 900 python bench.py --steps ${STEPS:-1000} --warmup 50 --workload $W > $OUT/${W}_bench.json 2> $OUT/${W}_bench.err; echo "bench $W exit=$?" >> $OUT/summary.txt
  rm -rf $OUT/prof_$W
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$W -o prof -- python bench.py --steps 60 --warmup 10 --streams 1 --workload $W --no-cpu-baseline --no-secondary --no-dist > $OUT/prof_$W.log 2>&1; echo "prof $W exit=$?" >> $OUT/summary.txt
  python scripts/frame_timeline.py $OUT/prof_$W/prof_kernel_trace.csv > $OUT/${W}_frame_timeline.txt 2>&1
  cp $OUT/prof_$W/prof_kernel_stats.csv $OUT/${W}_kernel_stats.csv 2>/dev/null
  find $OUT/prof_$W -name "*kernel_trace*" -size +4M -delete
done
for W in hd1m c3; do
  for O in 0 1; do
    WS_BLEND_ORDER=$O timeout 600 python scripts/blend_wait_breakdown.py $W 0 $OUT/blend_wait_breakdown_${W}_order$O.json > $OUT/blendwait_${W}_order$O.log 2>&1; echo "blendwait $W order=$O exit=$?" >> $OUT/summary.txt
  done
done
rm -f $OUT/*_raw.npz
# the driver's N > 1 form from a plain shell: two ranks on the one GPU
timeout 600 python bench.py --gpus 2 --single-device --dist-backend gloo --workload c2 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/selflaunch_2ranks.json 2> $OUT/selflaunch_2ranks.err; echo "selflaunch exit=$?" >> $OUT/summary.txt
# the default invocation as the driver runs it (hd1m headline + the c3 block), timed
( time timeout 600 python bench.py > $OUT/default_bench.json 2> $OUT/default_bench.err ) 2> $OUT/default_bench.time; echo "default bench exit=$?" >> $OUT/summary.txt
grep real $OUT/default_bench.time >> $OUT/summary.txt
# the driver's short form (20 steps) for the record
timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $OUT/default_bench_20steps.json 2>/dev/null
# four frames in flight under the tracer (in-flight kernel durations)
rm -rf $OUT/prof_inflight_hd1m
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_inflight_hd1m -o prof -- python bench.py --steps 300 --warmup 20 --workload hd1m --no-cpu-baseline --no-secondary --no-dist > $OUT/prof_inflight_hd1m.log 2>&1
cp $OUT/prof_inflight_hd1m/prof_kernel_stats.csv $OUT/hd1m_inflight_kernel_stats.csv 2>/dev/null
find $OUT/prof_inflight_hd1m -name "*kernel_trace*" -size +4M -delete
# frames in flight measured on the device (no tracer): who runs beside whom
timeout 600 python scripts/inflight_device_trace.py hd1m 1 2 3 4 5 6 8 > $OUT/inflight_device_trace_hd1m.jsonl 2> $OUT/inflight_device_trace_hd1m.err; echo "device trace hd1m exit=$?" >> $OUT/summary.txt
timeout 600 python scripts/inflight_device_trace.py c3 1 4 > $OUT/inflight_device_trace_c3.jsonl 2> $OUT/inflight_device_trace_c3.err; echo "device trace c3 exit=$?" >> $OUT/summary.txt
timeout 900 python scripts/measure_procedure.py $OUT/measure_rs_procedure.json > $OUT/measure.log 2>&1; echo "measure exit=$?" >> $OUT/summary.txt
for W in ${MORE_WORKLOADS:-c4 c5 c1 realistic1m}; do
  timeout 900 python bench.py --steps 600 --warmup 50 --workload $W > $OUT/${W}_bench.json 2> $OUT/${W}_bench.err; echo "bench $W exit=$?" >> $OUT/summary.txt
done
python - $OUT >> $OUT/summary.txt <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*_bench.json"))):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j["roofline"]
        print(os.path.basename(f), "fps", round(j["value"], 1), "single", round(j["config"].get("single_stream_fps", 0), 1), "roofline", r["kernel"],
              "frac", round(r["frac"], 3) if r["frac"] else None, "traffic", r.get("traffic"), "cpu", (j.get("cpu_baseline") or {}).get("value"),
              "enq_ms", round(j["config"].get("host_enqueue_ms_per_frame", 0), 4), "cores_busy", j["config"].get("host_cores_busy_per_rank"), "bound", r.get("bound"), "hbm_frac", round(r.get("hbm_frac") or 0, 3),
              "frame_frac", round((r.get("frame") or {}).get("frac") or 0, 3))
        if "secondary" in j:
            c3 = j["secondary"]["c3"]
            print("   secondary c3: fps", round(c3["value"], 1), "single", round(c3["single_stream_fps"], 1), "blend frac", c3["roofline"]["frac"],
                  {k: (round(v["ms_per_frame"], 4), round(v["frac"], 3)) for k, v in c3["kernels"].items()})
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
cat $OUT/summary.txt
# soak: frames in flight stay bit-identical over a long run (the digit width of the depth sort now changes from frame to frame)
for W in ${SOAK_WORKLOADS:-hd1m c5}; do
  timeout 900 python scripts/soak.py $W ${SOAK_ROUNDS:-300} $OUT/soak_$W.json > $OUT/soak_$W.log 2>&1; echo "soak $W exit=$? $(tail -1 $OUT/soak_$W.log | cut -c1-200)" >> $OUT/summary.txt
done
tail -8 $OUT/summary.txt
