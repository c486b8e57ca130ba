#!/bin/bash
# Which threads of a bench.py rank burn host CPU during the timed region, and where (GPU box; rocgdb user-space stacks)?
#   scripts/host_thread_probe.sh OUTDIR [bench args...]      (environment variables pass through to bench.py)
set -u
OUT=$1; shift
mkdir -p $OUT
MARK=/tmp/ws_bench_mark_$$
rm -f $MARK
WS_BENCH_MARK_FILE=$MARK python bench.py --steps ${PROBE_STEPS:-40000} --warmup 50 --no-secondary --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err &
BPID=$!
for i in $(seq 1 1200); do [[ -s $MARK ]] && break; sleep 0.1; done
PID=$(cat $MARK 2>/dev/null)
[[ -z "$PID" ]] && { echo "bench never reached its timed region" > $OUT/threads.txt; wait $BPID; exit 1; }
sleep 0.5
snap() { for t in /proc/$PID/task/*; do echo "$(basename $t) $(awk '{print $14+$15}' $t/stat 2>/dev/null) $(cat $t/comm 2>/dev/null) $(cat $t/wchan 2>/dev/null)"; done; }
snap > /tmp/s0_$$; sleep 1; snap > /tmp/s1_$$
join /tmp/s0_$$ /tmp/s1_$$ | awk '{d=$5-$2; if (d>0) print d, $1, $3, $7}' | sort -rn > $OUT/threads.txt   # ticks (10 ms) in 1 s, tid, comm, wchan
echo "pid $PID; busiest threads (ticks of 10 ms in 1 s, tid, comm, wchan):"; head -6 $OUT/threads.txt
for k in 1 2 3; do
  timeout 60 rocgdb -p $PID -batch -ex "thread apply all bt 14" > $OUT/stacks_$k.txt 2>&1
  sleep 0.3
done
for tid in $(head -3 $OUT/threads.txt | awk '{print $2}'); do
  echo "=== LWP $tid ==="
  for k in 1 2 3; do awk -v t="LWP $tid)" 'index($0,t){p=1;print;next} /^Thread /{p=0} p' $OUT/stacks_$k.txt | head -12; echo "--"; done
done > $OUT/busy_stacks.txt
cat $OUT/busy_stacks.txt | cut -c1-180 | head -120
wait $BPID
rm -f $MARK /tmp/s0_$$ /tmp/s1_$$
