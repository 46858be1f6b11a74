#!/bin/bash
# round 5, call O: the batch sizes where two pipelines' launches exceed the CUs by a few workgroups -- would the 80-sequence tile (fits) help?
set -o pipefail
O=gpurun_out/r05o; mkdir -p $O
Bn="--steps 8 --warmup 3 --no-cpu-baseline --no-other-configs --no-optional-modes --no-power"
one() { # label, args
  python bench.py $Bn $2 2> $O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],2))" || { tail -3 $O/err.txt; exit 1; }
}
for B in 32 33 34 35 36; do
  one "B=$B default" "--batch $B"
  one "B=$B inter2080" "--batch $B --seq-tile-inter 2080"
  one "B=$B inter2016" "--batch $B --seq-tile-inter 2016"
done
for B in 42 43 44 45 46; do
  one "B=$B default" "--batch $B"
  one "B=$B intra2080" "--batch $B --seq-tile-intra 2080"
done
