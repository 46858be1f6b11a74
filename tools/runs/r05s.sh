#!/bin/bash
# round 5, call S: ragged fold -- parity, then the full-length configuration with and without it
set -o pipefail
O=gpurun_out/r05s; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_inference.py -x -q -m gpu -k "ragged" > $O/ragged.txt 2>&1 || { tail -30 $O/ragged.txt; exit 1; }
tail -2 $O/ragged.txt
for f in off auto off auto; do
  DPRNN_LIN_FOLD=$f timeout -k 10 300 python tools/bench_fulllength.py 2>&1 | tail -2
done
