#!/bin/bash
out=gpurun_out/retune_win.log
: > $out
run() { echo -n "$* : " >> $out; timeout -k 10 200 python bench.py --extras none --no-cpu-baseline --steps 6 --warmup 2 "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms/step', round(d['ms_per_step'],3), 'windows', d['config']['l2_windows'])" >> $out; }
run
for kb in 4096 4608 5120 6144 6656 7168 8192; do run --window-kb $kb; done
for wc in 96 160 224 320; do run --tune win_cols=$wc; done
run --tune win_cols=160 --window-kb 7168
run --tune win_cols=224 --window-kb 9856
run --tune win_rpg=1
run
cat $out
