cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/t_m4 -o t -- python3 bench.py --extras none --no-cpu-baseline --steps 2 --warmup 1 --window-kb 3072 --tune sweep_mode=4,psync_pct=-1 > /dev/null 2>&1 || exit 3
find gpurun_out/t_m4 -name "*kernel_stats.csv" -exec grep -h "k_spmm\|k_hub" {} \; | sed 's/(.*)"//' | cut -c1-200
f=$(find gpurun_out/t_m4 -name "*kernel_trace.csv" | head -1); head -1 $f | cut -c1-400; grep "k_spmm_stream" $f | tail -6 | cut -c1-500
rm -rf gpurun_out/t_m4
