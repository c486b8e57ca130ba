#!/bin/bash
# WS_BLEND_COMPACT A/B (k_blend_c): tests first, then frames/s + blend kernel time per workload
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r04_${TAG:-compact}
mkdir -p $OUT
R=$OUT/compact.txt
: > $R
timeout 900 python -m pytest tests/test_gpu_render.py -m gpu -q -p no:cacheprovider -k "compaction or binning or cross_check or image_c" 2>&1 | tail -8 >> $R
for W in ${WORKLOADS:-hd1m c2 c3}; do
  for C in 0 1 0 1; do
    line=$(WS_BLEND_COMPACT=$C timeout 600 python bench.py --steps ${STEPS:-600} --warmup 30 --workload $W --no-cpu-baseline --no-secondary 2>$OUT/err_$W_$C.txt | tail -1)
    python -c "
import json,sys
try:
    j=json.loads(sys.argv[1]); k=j['kernels']['k_blend']
    print('$W compact=$C fps', round(j['value'],1), 'single', round(j['config']['single_stream_fps'],1), 'blend_us', round((k['avg_launch_ms'] or 0)*1e3,1), 'D', int(j['config']['avg_tile_entries']), j['config']['binning_tile'])
except Exception as e:
    print('$W compact=$C FAILED', e)" "$line" >> $R
  done
done
for C in 0 1; do
  rm -rf $OUT/prof_$C
  WS_BLEND_COMPACT=$C timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$C -o prof -- python bench.py --steps 60 --warmup 10 --streams 1 --workload hd1m --no-cpu-baseline --no-secondary --no-dist > $OUT/prof_$C.log 2>&1
  grep -E "k_blend" $OUT/prof_$C/prof_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120 >> $R
  cp $OUT/prof_$C/prof_kernel_stats.csv $OUT/hd1m_compact${C}_kernel_stats.csv 2>/dev/null
  find $OUT/prof_$C -name "*kernel_trace*" -size +4M -delete
done
cat $R
