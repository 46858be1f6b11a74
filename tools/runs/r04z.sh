#!/bin/bash
# Round 4, call Z: a small spill pipeline at the batch sizes where throughput steps down.
set -o pipefail
o=gpurun_out/r04z; mkdir -p $o
timeout -k 10 500 python tools/sweep_spill_group.py "20:10,10 21:10,11/10,10,1 22:11,11/10,10,2 23:11,12/10,10,3 32:16,16 33:16,17/16,16,1 34:17,17/16,16,2 35:17,18/16,16,3 37:18,19/18,18,1 42:21,21 43:21,22/21,21,1 44:22,22/21,21,2 46:23,23/21,21,4" > $o/spill_group.txt 2>&1; echo "rc=$?"; grep "^B=" $o/spill_group.txt | cut -c1-200
