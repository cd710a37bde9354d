cd $GRAFT_REPO_ROOT
for g in config community community-scrambled; do
  echo "== $g degree order"; timeout -k 10 200 python3 bench.py --workload products --graph $g --order degree --extras none --no-cpu-baseline --steps 5 | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['config']['nnz_per_rank'], d['config']['schedule'])" || exit 1
done
echo "== community natural order"; timeout -k 10 200 python3 bench.py --workload products --graph community --order degree --hub-threshold -2048 --hub-piece 8192 --extras none --no-cpu-baseline --steps 5 | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['config']['schedule'])"
echo "== config natural order"; timeout -k 10 200 python3 bench.py --workload products --graph config --order degree --hub-threshold -2048 --hub-piece 8192 --extras none --no-cpu-baseline --steps 5 | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['config']['schedule'])"
