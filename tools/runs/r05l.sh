#!/bin/bash
# round 5, call L: forward time by batch size (mode 3), Linear folded in (time-split) vs not, one box
set -o pipefail
O=gpurun_out/r05l; mkdir -p $O
DPRNN_LIN_FOLD=off timeout -k 10 500 python tools/latency_modes.py 1:64:1 3 > $O/sweep_off.txt 2>&1 || { tail -5 $O/sweep_off.txt; exit 1; }
DPRNN_LIN_FOLD=split timeout -k 10 500 python tools/latency_modes.py 1:64:1 3 > $O/sweep_split.txt 2>&1 || { tail -5 $O/sweep_split.txt; exit 1; }
paste -d'|' $O/sweep_off.txt $O/sweep_split.txt | sed 's/f16 pieces x3//g' | tail -64
for f in off split serial; do
  DPRNN_LIN_FOLD=$f timeout -k 10 300 python bench.py --batch 128 --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-optional-modes --no-power 2> $O/b128_$f.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B=128 $f', round(d['value'],1), round(d['ms_per_step'],2))"
done
