S32=/root/repo/web-splat_amd/lib_s32/libwebsplat_hip.so
cd /root/repo
# parity first (s32 lib)
WEBSPLAT_LIB=$S32 timeout 600 python -m pytest tests/test_gpu_render.py tests/test_gpu_fullsize.py tests/test_gpu_wgsl_golden.py -m gpu -q -p no:cacheprovider --maxfail 5 2>&1 | tail -4
for k in 1 2; do WHAT=abtest TAG=s32_$k VARIANTS="s20= s32=WEBSPLAT_LIB=$S32" WORKLOADS="hd1m c3 c2" bash scripts/gpu_r05.sh 2>&1 | grep -A1 "^ab " | cut -c1-330; done
# traffic of the blend with either stride on c3 and hd1m
for W in c3 hd1m; do
  for V in s20 s32; do
    L=""; [[ $V == s32 ]] && L=$S32
    WEBSPLAT_LIB=$L bash scripts/pmc_traffic.sh $W > /dev/null 2>&1
    python - $W $V <<'PY'
import json,sys
j=json.load(open(f"gpurun_out/traffic_{sys.argv[1]}.json"))
d=j["_detail"]
print("traffic",sys.argv[1],sys.argv[2],{k:(round(v["FETCH_SIZE_bytes_raw"]/1e6,1),round(v["WRITE_SIZE_bytes_raw"]/1e6,1)) for k,v in d.items() if k in ("k_blend","k_preprocess")})
PY
    cp gpurun_out/traffic_$W.json gpurun_out/traffic_${W}_$V.json
  done
done
