#!/bin/bash
# Round-3 bound experiments on the Reddit-shaped headline: ms per K=2 step for the product library and for side builds that
# make every gather hit (allhit), drop the y read-modify-write (normw), drop the gathers (nogather).  Results of the side
# builds are wrong by design: timing only.
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r3_exp_bounds.log
: > $out
for lib in "" dbg_allhit dbg_normw dbg_nogather dbg_allhit_normw; do
  for wkb in 5632 3584 2048; do
    if [ -n "$lib" ]; then export DISCO_HIP_LIB=$PWD/disco_amd/csrc/libdisco_$lib.so; else unset DISCO_HIP_LIB; fi
    ms=$(timeout -k 10 120 python3 -u bench.py --extras none --no-cpu-baseline --steps 3 --warmup 1 --window-kb $wkb "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), d['config']['l2_windows'])") || exit 1
    echo "lib=${lib:-product} window_kb=$wkb ms_per_step,windows: $ms" | tee -a $out
  done
done
