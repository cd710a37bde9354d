cd $GRAFT_REPO_ROOT
run() { echo "== $*"; timeout -k 10 200 python3 bench.py --workload products --extras none --no-cpu-baseline --steps 5 "$@" | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],3), d['roofline']['frac'])" || exit 1; }
run --graph community --hub-threshold -2048 --hub-piece 8192 --tune xcd_map=1
run --graph community --hub-threshold -2048 --hub-piece 8192
run --graph community --tune xcd_map=1
run --graph config --tune xcd_map=1
run --graph config --hub-threshold -2048 --hub-piece 8192 --tune xcd_map=1
