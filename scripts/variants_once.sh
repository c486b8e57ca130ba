cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/x; rm -f gpurun_out/x/*
for s in 1 2 6 8; do
  python bench.py --steps 400 --warmup 20 --no-cpu-baseline --streams $s > gpurun_out/x/c2_streams_$s.json 2> gpurun_out/x/c2_streams_$s.err
done
python bench.py --steps 300 --warmup 20 --no-cpu-baseline --workload hd1m > gpurun_out/x/hd1m.json 2>gpurun_out/x/hd1m.err
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --workload c3 > gpurun_out/x/c3.json 2>gpurun_out/x/c3.err
python bench.py --steps 300 --warmup 20 --no-cpu-baseline --workload c1 > gpurun_out/x/c1.json 2>gpurun_out/x/c1.err
