#!/bin/bash
# Profile one bench workload: timing run, kernel trace, and the three PMC passes (each in its own rocprofv3 run).
# Usage: tools/prof_workload.sh <tag> <bench.py args...>      outputs under gpurun_out/<tag>_*
tag="$1"; shift
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python3 -u bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
cat gpurun_out/${tag}_bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -o t -- python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/${tag}_trace.err || exit 2
for pass in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 300 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/${tag}_pmc_$name -o p -- python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/${tag}_pmc_$name.err || exit 3
done
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_TCC_HIT_sum --json gpurun_out/${tag}_l2.json | grep spmm
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_FETCH_SIZE --json gpurun_out/${tag}_fetch.json | grep spmm
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_WRITE_SIZE --json gpurun_out/${tag}_write.json | grep spmm
find gpurun_out/${tag}_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_kernel_stats.csv \;
# the raw per-dispatch files are large: keep the summaries only
rm -rf gpurun_out/${tag}_pmc_* gpurun_out/${tag}_trace
head -8 gpurun_out/${tag}_kernel_stats.csv
