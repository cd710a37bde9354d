#!/bin/bash
# leading dimension of the feature table: padded to 32 floats (default, rows are whole 128-B lines) or to 4 floats (16-B rows)
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r3_exp_ldpad.log
: > $out
for w in products reddit; do for pad in 32 4; do
  ms=$(timeout -k 10 150 python3 -u bench.py --workload $w --extras none --no-cpu-baseline --steps 6 --warmup 2 --ld-pad $pad 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), d['config']['ld'])") || exit 1
  echo "workload=$w ld_pad=$pad ms_per_step,ld: $ms" | tee -a $out
done; done
