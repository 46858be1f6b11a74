#!/bin/bash
# round 5, call P: where the 64-sequence tile (and the fold that comes with it) starts to beat the 16-sequence tile, per path
set -o pipefail
O=gpurun_out/r05p; mkdir -p $O
Bn="--steps 6 --warmup 3 --no-cpu-baseline --no-other-configs --no-optional-modes --no-power"
one() {
  python bench.py $Bn $2 2> $O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],2))" || { tail -3 $O/err.txt; exit 1; }
}
for B in 29 31 33 35 37 39 41 43 45 47 49 52 56 60 64; do
  one "B=$B default" "--batch $B"
  one "B=$B 16/16" "--batch $B --seq-tile-intra 2016 --seq-tile-inter 2016"
  one "B=$B 16/64" "--batch $B --seq-tile-intra 2016 --seq-tile-inter 2064"
  one "B=$B 64/64" "--batch $B --seq-tile-intra 2064 --seq-tile-inter 2064"
done
