#!/bin/bash
# WS_BLEND_PPL=2 (k_blend2: eight waves per 32x32 tile, two pixels per lane) against the default blend
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r04_${TAG:-ppl}
mkdir -p $OUT
R=$OUT/ppl.txt
: > $R
timeout 900 python -m pytest tests/test_gpu_render.py -m gpu -q -p no:cacheprovider -k "two_pixels or image_c" 2>&1 | tail -6 >> $R
for W in ${WORKLOADS:-hd1m c2 c3 c4}; do
  for P in 1 2 1 2; do
    line=$(WS_BLEND_PPL=$P timeout 600 python bench.py --steps ${STEPS:-600} --warmup 30 --workload $W --no-cpu-baseline --no-secondary 2>/dev/null | tail -1)
    python -c "
import json,sys
try:
    j=json.loads(sys.argv[1]); k=j['kernels']['k_blend']
    print('$W ppl=$P fps', round(j['value'],1), 'single', round(j['config']['single_stream_fps'],1), 'blend_us', round((k['avg_launch_ms'] or 0)*1e3,1), j['config']['binning_tile'])
except Exception as e:
    print('$W ppl=$P FAILED', e)" "$line" >> $R
  done
done
for P in 1 2; do
  rm -rf $OUT/prof_$P
  WS_BLEND_PPL=$P timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$P -o prof -- python bench.py --steps 60 --warmup 10 --streams 1 --workload hd1m --no-cpu-baseline --no-secondary --no-dist > $OUT/prof_$P.log 2>&1
  grep -E "k_blend" $OUT/prof_$P/prof_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120 >> $R
  cp $OUT/prof_$P/prof_kernel_stats.csv $OUT/hd1m_ppl${P}_kernel_stats.csv 2>/dev/null
  find $OUT/prof_$P -name "*kernel_trace*" -size +4M -delete
done
for P in 1 2 1 2; do
  line=$(WS_BLEND_PPL=$P python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1)
  python -c "
import json,sys
j=json.loads(sys.argv[1]); print('hd1m 20 steps ppl=$P fps', round(j['value'],1))" "$line" >> $R
done
cat $R
