#!/bin/bash
# Reddit-shaped headline: ms per K=2 step with the gathers issued through a buffer descriptor and cache-policy bits
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r3_exp_gather_policy.log
: > $out
for lib in "" dbg_aux0 dbg_aux1 dbg_aux2 dbg_aux16 dbg_aux17; do
  if [ -n "$lib" ]; then export DISCO_HIP_LIB=$PWD/disco_amd/csrc/libdisco_$lib.so; else unset DISCO_HIP_LIB; fi
  ms=$(timeout -k 10 120 python3 -u bench.py --extras none --no-cpu-baseline --steps 4 --warmup 1 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), d['config']['l2_windows'])") || exit 1
  echo "lib=${lib:-product} ms_per_step,windows: $ms" | tee -a $out
done
