#!/bin/bash
# products-shaped graph: hub threshold / piece grid (ms per step of K=2 hops), written to gpurun_out/r3_exp_hub.log
set -e
out=gpurun_out/r3_exp_hub.log
: > $out
for g in config config-truncated; do
for tp in "0 0" "512 1024" "512 2048" "2048 1024" "2048 2048" "2048 4096" "2048 8192" "8192 2048" "8192 8192" "32768 4096"; do
  set -- $tp
  timeout -k 10 120 python bench.py --workload products --graph $g --extras none --no-cpu-baseline --steps 8 --warmup 2 --hub-threshold $1 --hub-piece $2 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$g T=$1 P=$2', round(d['ms_per_step'],3), d['config']['schedule'])" >> $out
  tail -1 $out
done; done
