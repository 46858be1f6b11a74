#!/bin/bash
# Round 4, call U: where does the f16-pieces BPTT's step go?  Stand-alone (full grid, T = 200) with parts removed.
set -o pipefail
o=gpurun_out/r04u; mkdir -p $o
for v in default b1 b2 b3 b4 b8 b16 b32 b63; do
  if [ $v = default ]; then unset DPRNN_LIB; else export DPRNN_LIB=$PWD/tools/probes/diag/libdprnn_$v.so; fi
  echo "== $v" >> $o/bptt_probes.txt
  timeout -k 10 200 python tools/bench_recur.py --train 1 --gemm-mode 3 2>&1 | grep -E "f16 pieces|Error|error" >> $o/bptt_probes.txt || exit 1
done
cat $o/bptt_probes.txt
