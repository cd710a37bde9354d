#!/bin/bash
# In-flight depth (DISCO_U1 gathers per lane group) x occupancy (waves per SIMD the windowed kernel is compiled for):
# ms per K=2 step of the Reddit headline with side builds from tools/r3/build_unroll.py.
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r3_exp_unroll_occupancy.log
: > $out
for lib in "" dbg_u4_w8 dbg_u4_w6 dbg_u8_w5 dbg_u8_w4 dbg_u12_w5 dbg_u16_w4 dbg_u16_w3; do
  if [ -n "$lib" ]; then export DISCO_HIP_LIB=$PWD/disco_amd/csrc/libdisco_$lib.so; else unset DISCO_HIP_LIB; fi
  ms=$(timeout -k 10 120 python3 -u bench.py --extras none --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2))") || exit 1
  echo "lib=${lib:-product_u8_w6} ms_per_step: $ms" | tee -a $out
done
