#!/bin/bash
# write-through (sc1) output stores vs plain stores, per workload
out=gpurun_out/store_policy.log
: > $out
for v in "" _plainst; do
  lib=disco_amd/csrc/libdisco_hip$v.so
  for w in "reddit" "products" "products --graph community" "papers100M-shard" "arxiv" "reddit --sweep off"; do
    echo "== lib ${v:-sc1} workload $w" >> $out
    DISCO_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --workload $w --extras none --no-cpu-baseline --steps 5 --warmup 2 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   ms/step', round(d['ms_per_step'],3), 'windows', d['config']['l2_windows'])" >> $out || exit 1
  done
done
cat $out
