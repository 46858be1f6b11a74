#!/bin/bash
set -o pipefail
# deferred x' stores in the UNFOLDED inference kernels too (variant 2): kernel tests, then batch sizes that run unfolded (8, 16, 24) and 64
DPRNN_LIB=$PWD/tools/probes/diag/libdprnn_defer2.so timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "bilstm" 2>&1 | tail -2
Bn="--steps 12 --warmup 3 --no-cpu-baseline --no-other-configs --no-optional-modes --no-power"
for B in 8 16 24 36 64; do for i in 1 2; do for v in default defer2; do
  L=""; [ $v = defer2 ] && L=$PWD/tools/probes/diag/libdprnn_defer2.so
  DPRNN_LIB=$L python bench.py $Bn --batch $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B=$B $v', round(d['value'],1), round(d['ms_per_step'],3))"
done; done; done
