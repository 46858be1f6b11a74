#!/bin/bash
# round 5: training a narrow DPRNNTasNet on its zero-padded twin -- gradient goldens from the reference
set -e
mkdir -p gpurun_out/r05ff
timeout -k 10 500 python -m pytest tests/test_gpu_train.py -x -q -k "narrow or wide_small or wide_gln" -s > gpurun_out/r05ff/train_narrow.txt 2>&1 || { tail -n 40 gpurun_out/r05ff/train_narrow.txt; exit 1; }
grep -a "narrow\|passed\|failed" gpurun_out/r05ff/train_narrow.txt | tail -n 30
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "narrow" > gpurun_out/r05ff/parity_narrow.txt 2>&1 || { tail -n 40 gpurun_out/r05ff/parity_narrow.txt; exit 1; }
tail -n 3 gpurun_out/r05ff/parity_narrow.txt
