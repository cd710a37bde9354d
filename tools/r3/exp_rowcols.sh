#!/bin/bash
# products-shaped graph: column-block width of the plain row kernel (ms per step of K=2 hops)
set -e
out=gpurun_out/r3_exp_rowcols.log
: > $out
for rc in 0 32 64; do
  timeout -k 10 120 python bench.py --workload products --extras none --no-cpu-baseline --steps 8 --warmup 2 --tune row_cols=$rc 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('row_cols=$rc', round(d['ms_per_step'],3))" >> $out
  tail -1 $out
done
