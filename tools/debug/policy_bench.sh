#!/bin/bash
# windowed-kernel cache-policy variants (libdisco_hip_p?.so built with -DDISCO_WIN_* flags): Reddit default bench
out=gpurun_out/win_cache_policy.log
: > $out
for v in "" pA pB pC pD; do
  lib=disco_amd/csrc/libdisco_hip${v:+_$v}.so
  [ -f $lib ] || continue
  for wk in 0 4096 3072; do
    echo "== variant ${v:-base} window_kb=$wk" >> $out
    DISCO_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --extras none --no-cpu-baseline --steps 5 --warmup 2 --window-kb $wk 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   ms/step', round(d['ms_per_step'],3), 'windows', d['config']['l2_windows'])" >> $out || exit 1
  done
done
cat $out
