#!/bin/bash
export TMPDIR=/tmp
for n in 4 8; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/emul_trace$n -o t -- python3 bench.py --workload papers100M --emulate-world $n --extras none --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1 || exit 1
  echo "== N=$n"
  find gpurun_out/emul_trace$n -name "*kernel_stats.csv" -exec grep -E "k_spmm|k_hub" {} \; | cut -c1-60,200-400 | sed 's/([^)]*)//' | head -5
  rm -rf gpurun_out/emul_trace$n
done
