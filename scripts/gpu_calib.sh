#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (scripts/ubench/fetch_calib.*) -> gpurun_out/r03_calib/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_calib; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/ubench/fetch_calib.hip -o /tmp/fetch_calib || exit 1
/tmp/fetch_calib > $OUT/known.json
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- /tmp/fetch_calib > $OUT/pmc_$c.log 2>&1
done
python scripts/ubench/fetch_calib.py $OUT/known.json $OUT/pmc_FETCH_SIZE/pmc_counter_collection.csv $OUT/pmc_WRITE_SIZE/pmc_counter_collection.csv $OUT/fetch_calibration.json | tee $OUT/fetch_calibration.txt
find $OUT -name "*.csv" -size +2M -delete
