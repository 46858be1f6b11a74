#!/bin/bash
set -o pipefail
# correctness of the deferred x' stores first (the kernel tests through the variant library), then the A/B
DPRNN_LIB=$PWD/tools/probes/diag/libdprnn_lin_defer.so timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "time_split or direction_serial" 2>&1 | tail -2
DPRNN_LIB=$PWD/tools/probes/diag/libdprnn_lin_defer.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "folded_in or headline" 2>&1 | tail -2
Bn="--steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-optional-modes --no-power"
for i in 1 2 3; do for v in base defer; do
  DPRNN_LIB=$PWD/tools/probes/diag/libdprnn_lin_$v.so python bench.py $Bn 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],2))"
done; done
