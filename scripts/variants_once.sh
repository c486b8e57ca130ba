cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/x
for v in 0 1; do
  WS_BLEND_VARIANT=$v python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/x/blend_v$v.json 2> gpurun_out/x/blend_v$v.err
done
for s in 2 3 4; do
  python bench.py --steps 300 --warmup 20 --no-cpu-baseline --streams $s > gpurun_out/x/streams_$s.json 2> gpurun_out/x/streams_$s.err
done
WS_BLEND_VARIANT=1 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --streams 2 > gpurun_out/x/v1_streams_2.json 2>&1
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload hd1m > gpurun_out/x/hd1m.json 2>&1
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --workload c3 > gpurun_out/x/c3.json 2>&1
