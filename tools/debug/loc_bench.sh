#!/bin/bash
# products-shaped community graph behind a random relabelling: degree order vs locality pass (+ XCD-aware block order)
set -e
out=gpurun_out/loc_bench.log
: > $out
run() { echo "== $*" >> $out; timeout -k 10 200 python bench.py --workload products --extras none --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>&1 | tail -1 >> $out; }
G="--graph community-scrambled --order locality"
run $G --locality-max-group 0
run $G
run $G --tune xcd_map=1
run $G --tune xcd_map=128
run $G --tune xcd_map=512
run $G --tune xcd_map=1024
run $G --locality-max-group 2048 --tune xcd_map=256
run $G --locality-max-group 32768 --tune xcd_map=1024
run $G --locality-sweeps 16 --tune xcd_map=512
run --graph community --order locality --tune xcd_map=512
run --graph community --order degree
run --graph config --order locality --tune xcd_map=512
