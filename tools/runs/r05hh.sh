#!/bin/bash
# round 5: does a second resident BPTT workgroup per CU overlap phases?  (occupancy probe)
set -e
mkdir -p gpurun_out/r05hh
timeout -k 10 300 python tools/bench_recur.py --train 3 --T 200 > gpurun_out/r05hh/bptt_occupancy.txt 2>&1 || { tail -n 30 gpurun_out/r05hh/bptt_occupancy.txt; exit 1; }
cat gpurun_out/r05hh/bptt_occupancy.txt
