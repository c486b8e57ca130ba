cd "${GRAFT_REPO_ROOT:-/root/repo}"
L=web-splat_amd
for cfg in "1000000 3840 2160" "5000000 3840 2160" "250000 3840 2160"; do
 for v in "base:" "mw8t1:WS_BLEND_TPW_LOG2=1 WEBSPLAT_LIB=$L/lib_mw8/libwebsplat_hip.so" "mw8t2:WS_BLEND_TPW_LOG2=2 WEBSPLAT_LIB=$L/lib_mw8/libwebsplat_hip.so"; do
  n=${v%%:*}; e=${v#*:}
  echo "== $cfg $n: $(env $e NONNULL=1 timeout 300 python scripts/sweep_one.py $cfg 4 1 2>&1 | grep "frames.s" | tr '\n' ' ')"
 done
done
