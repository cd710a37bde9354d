#!/bin/bash
# Does `nt` on the gathers change what the L2 retains?  products-shaped configuration model, plain kernel.
export TMPDIR=/tmp
out=gpurun_out/gather_nt.log
echo "# k_spmm_rows with every gather as global_load ... nt (libdisco_hip_gnt.so, -DDISCO_GATHER_NT=1) vs plain loads; products config, reddit --sweep off" > $out
for v in "" _gnt; do
  for w in "products" "reddit --sweep off" "reddit"; do
    echo "== lib ${v:-plain} workload $w" >> $out
    DISCO_HIP_LIB=$PWD/disco_amd/csrc/libdisco_hip$v.so timeout -k 10 200 python3 bench.py --workload $w --extras none --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   ms/step', round(d['ms_per_step'],3))" >> $out || exit 1
  done
  DISCO_HIP_LIB=$PWD/disco_amd/csrc/libdisco_hip$v.so timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/gnt_pmc$v -o p -- python3 bench.py --workload products --extras none --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1 || exit 2
  echo "== lib ${v:-plain} products: L2 counters" >> $out
  python3 tools/pmc_summary.py gpurun_out/gnt_pmc$v | grep spmm >> $out
  rm -rf gpurun_out/gnt_pmc$v
done
cat $out
