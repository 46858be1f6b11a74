#!/bin/bash
# round 5, call R: the two batch sizes that still drop by more than 2 % against B - 1 (21, 33): pipeline sizes / order / count
set -o pipefail
O=gpurun_out/r05r; mkdir -p $O
Bn="--steps 8 --warmup 3 --no-cpu-baseline --no-other-configs --no-optional-modes --no-power"
one() {
  python bench.py $Bn $2 2> $O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],2))" || { tail -3 $O/err.txt; exit 1; }
}
one "B=20 default" "--batch 20"
one "B=21 default(10,11)" "--batch 21"
one "B=21 11,10" "--batch 21 --group-sizes 11,10"
one "B=21 7,7,7" "--batch 21 --group-sizes 7,7,7"
one "B=21 8,8,5" "--batch 21 --group-sizes 8,8,5"
one "B=21 6,5,5,5" "--batch 21 --group-sizes 6,5,5,5"
one "B=21 one pipeline" "--batch 21 --group-sizes 21"
one "B=22 default" "--batch 22"
one "B=32 default" "--batch 32"
one "B=33 default(16,17)" "--batch 33"
one "B=33 17,16" "--batch 33 --group-sizes 17,16"
one "B=33 11,11,11" "--batch 33 --group-sizes 11,11,11"
one "B=33 16,16,1" "--batch 33 --group-sizes 16,16,1"
one "B=33 9,8,8,8" "--batch 33 --group-sizes 9,8,8,8"
one "B=34 default" "--batch 34"
