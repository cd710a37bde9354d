cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
pm() { tag=$1; shift; timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/x_$tag -o p -- python3 bench.py --extras none --no-cpu-baseline --steps 2 --warmup 1 "$@" > /dev/null 2>&1 || return 1; echo "== $tag"; python3 tools/pmc_summary.py gpurun_out/x_$tag | grep "spmm_stream" | cut -c1-300; rm -rf gpurun_out/x_$tag; }
pm paced1 --window-kb 3072 --tune sweep_mode=4,psync_pct=1 &&
pm paced4 --window-kb 1536 --tune sweep_mode=4,psync_pct=4 &&
pm unpaced --window-kb 3072 --tune sweep_mode=4,psync_pct=-1
