#!/bin/bash
set -o pipefail
O=gpurun_out/r05ab; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_inference.py -x -q -k "ragged" > $O/ragged_tests.txt 2>&1 || { tail -n 40 $O/ragged_tests.txt; exit 1; }
tail -n 2 $O/ragged_tests.txt
for v in 64 1 64 1; do
  DPRNN_RAGGED_AUX_BATCH=$v FL_SHARE=1 timeout -k 10 300 python tools/bench_fulllength.py ira 2>&1 | grep -a "ragged batch\|whole share (\|worst rel" | head -n 3 | sed "s/^/speaker passes batched $v: /" | cut -c1-210 | tee -a $O/ab.txt
done
