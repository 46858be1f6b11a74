#!/bin/bash
set -o pipefail
O=gpurun_out/r05v; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_inference.py -x -q -m gpu > $O/inference.txt 2>&1 || { tail -30 $O/inference.txt; exit 1; }
tail -2 $O/inference.txt
for p in 1 2 1 2 3; do
  DPRNN_RAGGED_PIPELINES=$p timeout -k 10 300 python tools/bench_fulllength.py 2>&1 | grep "whole share (375" | sed "s/^/pipelines=$p: /"
done
