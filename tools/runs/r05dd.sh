#!/bin/bash
set -o pipefail
Bn="--steps 8 --warmup 3 --no-cpu-baseline --no-other-configs --no-optional-modes --no-power"
one() { env $3 python bench.py $Bn $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],2))"; }
for B in 24 34 38 42 46; do
  one "B=$B default" "--batch $B" ""
  one "B=$B 32/32 folded" "--batch $B --seq-tile-intra 2032 --seq-tile-inter 2032" "DPRNN_LIN_FOLD=split"
  one "B=$B 32/32 unfolded" "--batch $B --seq-tile-intra 2032 --seq-tile-inter 2032" "DPRNN_LIN_FOLD=off"
  one "B=$B 16/32 folded" "--batch $B --seq-tile-intra 2016 --seq-tile-inter 2032" "DPRNN_LIN_FOLD=split"
  one "B=$B 32/16 folded" "--batch $B --seq-tile-intra 2032 --seq-tile-inter 2016" "DPRNN_LIN_FOLD=split"
done
