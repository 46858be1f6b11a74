#!/bin/bash
set -o pipefail
Bn="--steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-optional-modes --no-power"
for i in 1 2; do for v in base prio0 prio2 seprcp; do
  DPRNN_LIB=$PWD/tools/probes/diag/libdprnn_lin_$v.so python bench.py $Bn 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],2))"
done; done
