#!/bin/bash
# round 5: exact-fp32 forward at B = 64 under other pipeline counts (does packing recover more of the 0.84-0.88 a full grid reaches?)
set -o pipefail
O=gpurun_out/r05ss; mkdir -p $O
Bn="--gemm-mode 0 --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-optional-modes --no-power"
one() { timeout -k 10 200 python bench.py $Bn $2 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), 'utt/s', round(d['ms_per_step'],1), 'ms')" | tee -a $O/exact_groups.txt; }
one "default (4 x 16)" ""
one "2 x 32" "--group-sizes 32,32"
one "3 (22,21,21)" "--group-sizes 22,21,21"
one "5 (13,13,13,13,12)" "--group-sizes 13,13,13,13,12"
one "6 (11,11,11,11,10,10)" "--group-sizes 11,11,11,11,10,10"
one "8 x 8" "--group-sizes 8,8,8,8,8,8,8,8"
one "default again" ""
