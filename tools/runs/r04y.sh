#!/bin/bash
# Round 4, call Y: batch sweep with the uniform-tile policy and the wide-tile margin; parity tests of the forward at a few sizes.
set -o pipefail
o=gpurun_out/r04y; mkdir -p $o
timeout -k 10 600 python tools/latency_modes.py 8:44:1,48:64:4 3 > $o/batch_sweep_mode3.txt 2>&1; echo "rc=$?"; grep "^B=" $o/batch_sweep_mode3.txt | cut -c1-120
timeout -k 10 600 python -m pytest tests/test_gpu_inference.py tests/test_gpu_parity.py -x -q -m gpu > $o/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $o/tests.txt
