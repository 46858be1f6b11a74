#!/bin/bash
# round 5: round-over-round on ONE box -- the round-4 final tree (git worktree of 5a979de, its own library) against this tree, alternating
set -o pipefail
O=gpurun_out/r05nn; mkdir -p $O
Bn="--steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-optional-modes --no-power"
one() { ( cd $2 && timeout -k 10 300 python bench.py $Bn 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), 'utt/s', round(d['ms_per_step'],2), 'ms')" ); }
for i in 1 2 3; do
  one "round-4 tree" _r04tree | tee -a $O/ab.txt
  one "round-5 tree" . | tee -a $O/ab.txt
done
Bt="--mode train --no-power"
( cd _r04tree && timeout -k 10 300 python bench.py $Bt 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('round-4 tree, training step', round(d['ms_per_step'],1), 'ms')" ) | tee -a $O/ab.txt
timeout -k 10 300 python bench.py $Bt 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('round-5 tree, training step', round(d['ms_per_step'],1), 'ms')" | tee -a $O/ab.txt
