#!/bin/bash
# Round 4, call D: adaptive activation scale (kernel + end-to-end tests, whole kernel / parity files), where the graph memory sits,
# the graph-churn arms that separate "segments freed during a capture" from "a graph destroyed between another's instantiation and
# first launch", the batch-size sweep before the tile / group rework.
set -o pipefail
o=gpurun_out/r04d; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q -m gpu > $o/tests.txt 2>&1 || { tail -60 $o/tests.txt; exit 1; }
tail -3 $o/tests.txt
timeout -k 10 200 python tools/probes/graph_memory.py 64 > $o/graph_memory.txt 2>&1; cat $o/graph_memory.txt | head -30
timeout -k 10 500 python tools/latency_modes.py 1:32:1,36:64:4 3,0 > $o/batch_sweep_before.txt 2>&1; cat $o/batch_sweep_before.txt
CHURN_LEGACY=2 timeout -k 10 400 python tools/probes/graph_churn.py 4 16,24,32,48,64,96,128 > $o/churn_legacy_evict_first.txt 2>&1; tail -6 $o/churn_legacy_evict_first.txt | cut -c1-420
CHURN_LEGACY=1 timeout -k 10 400 python tools/probes/graph_churn.py 4 16,24,32,48,64,96,128 > $o/churn_legacy.txt 2>&1; tail -12 $o/churn_legacy.txt | cut -c1-420
true
