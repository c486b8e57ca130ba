cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/x; rm -f gpurun_out/x/*
for s in 2 4 6 8 12; do
  python bench.py --steps 600 --warmup 30 --no-cpu-baseline --streams $s > gpurun_out/x/c2_streams_$s.json 2> gpurun_out/x/c2_streams_$s.err
done
for s in 4 8; do
python bench.py --steps 400 --warmup 20 --no-cpu-baseline --workload hd1m --streams $s > gpurun_out/x/hd1m_$s.json 2>gpurun_out/x/hd1m_$s.err
done
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --workload c3 > gpurun_out/x/c3.json 2>gpurun_out/x/c3.err
