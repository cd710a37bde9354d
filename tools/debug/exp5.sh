cd $GRAFT_REPO_ROOT
run() { echo "== $*"; timeout -k 10 200 python3 bench.py --extras none --no-cpu-baseline --steps 5 "$@" | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],3), d['config']['l2_windows'], d['roofline']['avg_launch_ms'])" || exit 1; }
run
for wkb in 1024 1536 2048 3072; do run --window-kb $wkb --tune sweep_mode=4; done
run --window-kb 1536 --tune sweep_mode=4,psync_pct=-1
run --window-kb 2048 --tune sweep_mode=4,psync_polls=8
