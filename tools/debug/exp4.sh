cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for wkb in 5632 3072; do
  echo "== window ${wkb} KiB"
  timeout -k 10 200 python3 bench.py --extras none --no-cpu-baseline --steps 5 --window-kb $wkb | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],3), d['config']['l2_windows'])" || exit 1
  timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/x_$wkb -o p -- python3 bench.py --extras none --no-cpu-baseline --steps 2 --warmup 1 --window-kb $wkb > /dev/null 2>&1 || exit 2
  python3 tools/pmc_summary.py gpurun_out/x_$wkb | grep "spmm" | cut -c1-260; rm -rf gpurun_out/x_$wkb
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/t_$wkb -o t -- python3 bench.py --extras none --no-cpu-baseline --steps 2 --warmup 1 --window-kb $wkb > /dev/null 2>&1 || exit 3
  find gpurun_out/t_$wkb -name "*kernel_stats.csv" -exec grep -h "k_spmm" {} \; | sed 's/(.*)"//' | cut -c1-200; rm -rf gpurun_out/t_$wkb
done
