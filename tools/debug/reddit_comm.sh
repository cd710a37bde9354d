#!/bin/bash
out=gpurun_out/reddit_comm.log
: > $out
run() { echo "== $*" >> $out; timeout -k 10 200 python bench.py --extras none --no-cpu-baseline --steps 5 --warmup 2 "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   ms/step', round(d['ms_per_step'],3), 'windows', d['config']['l2_windows'], d['config']['row_order'], d['config'].get('locality_pass'))" >> $out; }
run --workload reddit --graph community
run --workload reddit --graph community-scrambled
run --workload reddit --graph config
run --workload reddit --graph community --order locality --sweep off
run --workload reddit --graph community-scrambled --order locality --sweep off
run --workload reddit --graph community --order degree --sweep on
run --workload products --graph community
run --workload products --graph community-scrambled
run --workload papers100M-shard --graph community
cat $out
