#!/bin/bash
set -o pipefail
O=gpurun_out/r05zz; mkdir -p $O
for v in 8 16 32 8 16 32; do
  for q in 16; do
  HWQ=$q DPRNN_RAGGED_STREAMS=$v FL_SHARE=1 timeout -k 10 300 python tools/bench_fulllength.py ira 2>&1 | grep -a "whole share (" | head -n 1 | sed "s/^/side streams $v (hw queues $q): /" | cut -c1-200 | tee -a $O/ab.txt
  done
done
HWQ=32 DPRNN_RAGGED_STREAMS=32 FL_SHARE=1 timeout -k 10 300 python tools/bench_fulllength.py ira 2>&1 | grep -a "whole share (" | head -n 1 | sed "s/^/side streams 32 (hw queues 32): /" | cut -c1-200 | tee -a $O/ab.txt
