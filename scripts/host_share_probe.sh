#!/bin/bash
# What does a rank of an 8-GPU run need from the host?  bench.py's hd1m line (4 frames in flight) under different shares of
# the box's CPUs, with and without the other ranks' stand-ins (scripts/ubench/cpu_burn.c) -> gpurun_out/r04_<TAG>/host_share.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04_${TAG:-host}
mkdir -p $OUT
R=$OUT/host_share.txt
: > $R
{
  echo "nproc $(nproc)"; echo "cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; echo "cpuset $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null)"
  grep Cpus_allowed_list /proc/self/status
  lscpu | grep -E "^CPU\(s\)|Thread|Core|Socket|NUMA node[0-9]"
  echo "cpu0 siblings $(cat /sys/devices/system/cpu/cpu0/topology/thread_siblings_list 2>/dev/null) core_id $(cat /sys/devices/system/cpu/cpu0/topology/core_id 2>/dev/null)"
  rocm-smi --showtopo 2>/dev/null | grep -iE "numa|GPU\[0\]" | head -6
} >> $R 2>&1
gcc -O2 -pthread scripts/ubench/cpu_burn.c -o /tmp/cpu_burn
MINE=$(python - <<'PY'
import os, sys
sys.path.insert(0, ".")
import bench
cores = bench._physical_cores(sorted(os.sched_getaffinity(0)))
k = len(cores) // 8
print(",".join(str(c) for g in cores[:k] for c in g))
PY
)
OTHERS=$(python - <<'PY'
import os, sys
sys.path.insert(0, ".")
import bench
cores = bench._physical_cores(sorted(os.sched_getaffinity(0)))
k = len(cores) // 8
print(" ".join(str(c) for g in cores[k:] for c in g))
PY
)
echo "mine: $MINE" >> $R
one() {  # label, prefix command..., env via ENVV
  local label=$1; shift
  local line
  line=$(env ${ENVV:-} "$@" python bench.py --steps ${STEPS:-1000} --warmup 50 --no-cpu-baseline --no-secondary ${BARGS:-} 2> $OUT/host_$label.err | tail -1)
  python - "$label" "$line" >> $R <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2]); c = j["config"]
    print(f"{sys.argv[1]:34s} fps {j['value']:8.1f}  single {c['single_stream_fps']:7.1f}  enqueue_ms/frame {c['host_enqueue_ms_per_frame']:.4f}  host_bound {c['host_bound']}  cpus {c['host_cpus']}")
except Exception as e:
    print(f"{sys.argv[1]:34s} FAILED {e}")
PY
}
one free
one taskset_eighth taskset -c $MINE
one taskset_4cpus taskset -c $(echo $MINE | cut -d, -f1-4)
/tmp/cpu_burn 900 $OTHERS > /dev/null &
BP=$!
sleep 1
one free_with_burners
one eighth_with_burners taskset -c $MINE
ENVV="WS_GRAPH=1" one eighth_with_burners_graph taskset -c $MINE
ENVV="HIP_FORCE_DEV_KERNARG=1" one eighth_with_burners_devkernarg taskset -c $MINE
ENVV="GPU_MAX_HW_QUEUES=4" one eighth_with_burners_4queues taskset -c $MINE
BARGS="--streams 1" one eighth_with_burners_1stream taskset -c $MINE
BARGS="--no-dist" one eighth_with_burners_nodist taskset -c $MINE
kill $BP; wait $BP 2>/dev/null
# burners on HALF of the other CPUs only (is it the sheer number of busy CPUs, or the SMT siblings / same-CCX neighbours?)
HALF=$(echo $OTHERS | tr ' ' '\n' | awk 'NR%2==0' | tr '\n' ' ')
/tmp/cpu_burn 600 $HALF > /dev/null &
BP=$!
sleep 1
one eighth_with_half_burners taskset -c $MINE
kill $BP; wait $BP 2>/dev/null
cat $R
