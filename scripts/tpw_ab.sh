cd "${GRAFT_REPO_ROOT:-/root/repo}"
for w in c2 hd1m c5; do
 for t in 0 1 2; do
  WS_BLEND_TPW_LOG2=$t python bench.py --steps 300 --warmup 20 --workload $w --views 16 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$w tpw_log2=$t fps', round(j['value'],1), 'single', round(j['config']['single_stream_fps'],1), 'blend us', round(j['kernels']['k_blend']['avg_launch_ms']*1e3,1))"
 done
done
