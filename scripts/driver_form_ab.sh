#!/bin/bash
# Same-box A/B of the bench line: the round-4 tree (.r04ref: a worktree of 518f592 with its own library) against this tree under
# host-side variants; 1000-step lines and the driver's 20-step form, alternating.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_driverform${TAG:-}; mkdir -p $OUT; rm -f $OUT/summary.txt
val() { python -c "
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=j['config']
k=j.get('kernels',{})
print(round(j['value'],1), 'single', round(c.get('single_stream_fps',0)), 'busy', round(c.get('host_cores_busy_per_rank',0),2), 'intervals', {a:round(b['event_interval_ms']*1e3,1) for a,b in k.items() if a in ('k_preprocess','depth:k_sort_col_scan','k_bin_emit','tiles:k_sort_scatter','k_blend')})" $1 2>&1; }
run() {  # name dir steps [env...] -- [args...]
  local N=$1 D=$2 S=$3; shift 3
  local envs=(); while [[ $# -gt 0 && $1 != -- ]]; do envs+=("$1"); shift; done; shift
  ( cd $D && env "${envs[@]}" timeout 300 python bench.py --gpus 1 --steps $S --warmup $([[ $S == 20 ]] && echo 5 || echo 50) --no-secondary --no-cpu-baseline "$@" ) > $OUT/${N}_$S.json 2> $OUT/${N}_$S.err
  echo "$N steps$S: $(val $OUT/${N}_$S.json)" >> $OUT/summary.txt
}
for k in 1 2 3; do
  for S in 1000 20; do
    run r04_$k .r04ref $S X=1 --
    run r05_block_$k . $S X=1 -- --host-wait block
    run r05_spin_$k . $S X=1 -- --host-wait spin
    [[ -n ${MORE:-} ]] && run r05_spin_nowin_$k . $S WS_BATCH_QUEUE_DEPTH=0 -- --host-wait spin
    [[ -n ${MORE:-} ]] && run r05_spin_noskip_$k . $S WS_DEPTH_SKIP_TOP=0 -- --host-wait spin
  done
done
cat $OUT/summary.txt
