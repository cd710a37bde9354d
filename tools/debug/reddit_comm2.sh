#!/bin/bash
out=gpurun_out/reddit_comm2.log
: > $out
run() { echo -n "$* : " >> $out; timeout -k 10 200 python bench.py --workload reddit --graph community --extras none --no-cpu-baseline --steps 5 --warmup 2 "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms/step', round(d['ms_per_step'],3), 'windows', d['config']['l2_windows'], d['config']['row_order'])" >> $out; }
run
for s in 32 64 256 512 1024; do run --tune xcd_map=$s; done
for c in 64 96 160 256; do run --tune row_cols=$c; done
run --tune row_cols=64,xcd_map=256
run --tune row_cols=64,xcd_map=512
run --locality-max-group 4096 --tune xcd_map=512
run --locality-max-group 2048 --tune xcd_map=256
cat $out
