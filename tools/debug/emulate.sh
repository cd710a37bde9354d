#!/bin/bash
# What ONE rank of an N-rank feature-partition job does (no exchange: this IS the job's step time), measured on one GPU.
out=gpurun_out/feature_partition_emulated2.log; : > $out

for w in "papers100M" "papers100M --graph community"; do
  for n in 2 4 8; do
    echo "== $w  N=$n" >> $out
    timeout -k 10 400 python bench.py --workload $w --emulate-world $n --extras none --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); n=$n
print('   ms/step', round(d['ms_per_step'],2), ' job edges*feat/s %.3e' % (d['value']*max(n,1)), ' frac', d['roofline']['frac'], ' setup_s', d['config']['setup_s'], ' windows', d['config']['l2_windows'], d['config']['row_order'])" >> $out || exit 1
  done
done
cat $out
