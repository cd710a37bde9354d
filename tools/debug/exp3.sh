cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
pm() { tag=$1; shift; timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/x_$tag -o p -- python3 bench.py --workload products --extras none --no-cpu-baseline --steps 2 --warmup 1 "$@" > /dev/null 2>&1 || return 1; echo "== $tag"; python3 tools/pmc_summary.py gpurun_out/x_$tag | grep "spmm_rows" | cut -c1-300; rm -rf gpurun_out/x_$tag; }
pm comm_nat_xcd --graph community --hub-threshold -2048 --hub-piece 8192 --tune xcd_map=1 &&
pm comm_nat --graph community --hub-threshold -2048 --hub-piece 8192
