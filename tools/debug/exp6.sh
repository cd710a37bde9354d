cd $GRAFT_REPO_ROOT
run() { echo "== $*"; timeout -k 10 200 python3 bench.py --extras none --no-cpu-baseline --steps 3 "$@" | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],3), d['config']['l2_windows'], d['roofline']['avg_launch_ms'])" || exit 1; }
for wkb in 1536 3072 6144 12288; do run --window-kb $wkb --tune sweep_mode=4,psync_pct=-1; done
run --window-kb 3072 --tune sweep_mode=4,psync_pct=-1,stream_u=12
run --window-kb 3072 --tune sweep_mode=4,psync_pct=-1,stream_u=16
run --window-kb 3072 --tune sweep_mode=4,psync_pct=-1,sweep_blocks_per_cu=2
