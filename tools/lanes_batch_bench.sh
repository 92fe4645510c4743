#!/bin/bash
# bench.py at other lane counts / batch sizes (the headline is lanes 4, batch 256).  GPU box: bash tools/lanes_batch_bench.sh
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(sys.argv[1], round(d['value']), d['step_ms']['median'])" "$1"; }
B="--steps 20 --warmup 5 --cpu-seconds 0 --no-detect"
for l in 3 4 5 6 4; do python3 bench.py $B --lanes $l 2>/dev/null | line "lanes=$l"; done
for b in 128 512 1024; do python3 bench.py $B --batch $b 2>/dev/null | line "batch=$b"; done
