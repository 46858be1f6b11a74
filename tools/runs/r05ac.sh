#!/bin/bash
set -o pipefail
O=gpurun_out/r05ac; mkdir -p $O
DPRNN_RAGGED_TRACE=1 FL_ONLY=share timeout -k 10 300 python tools/bench_fulllength.py ira > $O/host_trace.txt 2>&1 || { tail -n 20 $O/host_trace.txt; exit 1; }
grep -a "ragged host\|whole share" $O/host_trace.txt | tail -n 22
