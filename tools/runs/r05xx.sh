#!/bin/bash
set -o pipefail
O=gpurun_out/r05yy; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_inference.py -x -q -k "ragged" > $O/ragged_tests.txt 2>&1 || { tail -n 40 $O/ragged_tests.txt; exit 1; }
tail -n 2 $O/ragged_tests.txt
for v in 1 0 1 0; do
  DPRNN_RAGGED_BN_FOLDS_ONCE=$v FL_SHARE=1 timeout -k 10 300 python tools/bench_fulllength.py ira 2>&1 | grep -a "ragged batch\|whole share (\|worst rel" | head -n 3 | sed "s/^/bn folds once $v: /" | tee -a $O/ab.txt
done
