#!/bin/bash
# Round 4, call L: batch sweep with the fractional-round tile choice (mode 3), B = 20 .. 64
set -o pipefail
o=gpurun_out/r04l2; mkdir -p $o
timeout -k 10 600 python tools/latency_modes.py 8:32:1,36:64:4 3 > $o/batch_sweep_frac.txt 2>&1; cat $o/batch_sweep_frac.txt
