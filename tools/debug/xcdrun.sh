#!/bin/bash
out=gpurun_out/xcdrun2.log
: > $out
run() { echo -n "$* : " >> $out; timeout -k 10 300 python bench.py --extras none --no-cpu-baseline --steps 5 --warmup 2 "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms/step', round(d['ms_per_step'],3), 'xcd_run', d['config'].get('xcd_run'), d['config']['row_order'], (d['config'].get('locality_pass') or {}).get('edges_inside_a_group'))" >> $out; }
for g in 2048 4096 16384; do
run --workload products --graph community --locality-max-group $g
run --workload products --graph community-scrambled --locality-max-group $g
run --workload papers100M-shard --graph community --locality-max-group $g
run --workload papers100M-shard --graph community-scrambled --locality-max-group $g
done
cat $out
