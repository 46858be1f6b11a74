#!/bin/bash
# Round 4, call S: what do the training forward's stores cost?  fwd_train / BPTT stand-alone (full grid, T = 200) with the tape stores
# (probe 256) and the h stores (probe 512) removed, against the inference kernel on the same grid.
set -o pipefail
o=gpurun_out/r04s; mkdir -p $o
for v in default d256 d512 d768; do
  if [ $v = default ]; then unset DPRNN_LIB; else export DPRNN_LIB=$PWD/tools/probes/diag/libdprnn_$v.so; fi
  echo "== $v" >> $o/train_probes.txt
  timeout -k 10 200 python tools/bench_recur.py --train 1 --gemm-mode 3 2>&1 | grep -E "train|Error|error" >> $o/train_probes.txt || exit 1
done
cat $o/train_probes.txt
