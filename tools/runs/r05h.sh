#!/bin/bash
# round 5, call H: what a step of the time-split phases costs -- probe builds with one part removed each (results wrong, timing valid)
set -o pipefail
O=gpurun_out/r05h; mkdir -p $O
for v in base d64 d1024 d3072 d16 d1 d8 d4; do
  echo "== $v" >> $O/diag.txt
  DPRNN_LIB=$PWD/tools/probes/diag/libdprnn_lin_$v.so timeout -k 10 200 python tools/bench_serial.py --batch 32 2>&1 | grep "time-split" >> $O/diag.txt || exit 1
done
cat $O/diag.txt
