#!/bin/bash
set -o pipefail
O=gpurun_out/r05t; mkdir -p $O
Bn="--steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-optional-modes --no-power"
one() {
  python bench.py $Bn $2 2> $O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],2))" || { tail -3 $O/err.txt; exit 1; }
}
for i in 1 2; do
one "B=64 default (32,32)" ""
one "B=64 22,21,21" "--group-sizes 22,21,21"
one "B=64 16x4" "--group-sizes 16,16,16,16"
one "B=64 40,24" "--group-sizes 40,24"
one "B=64 one" "--group-sizes 64"
done
