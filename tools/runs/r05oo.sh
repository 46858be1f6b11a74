#!/bin/bash
set -e
mkdir -p gpurun_out/r05oo
for v in gn1 gn2 gn4 gn1 gn2 gn4; do
  DPRNN_LIB=$PWD/tools/probes/diag/libdprnn_$v.so timeout -k 10 120 python tools/probes/gn_bwd_rows.py 2>&1 | grep -a "^lib" | tee -a gpurun_out/r05oo/gn.txt
done
timeout -k 10 300 python -m pytest tests/test_gpu_train_kernels.py -x -q -k "gn or norm or group" 2>&1 | tail -n 3
