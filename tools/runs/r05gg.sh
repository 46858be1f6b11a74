#!/bin/bash
# round 5: narrow Spe / IRA models train on their twins; Adam trajectory on a narrow model with the fused optimizer
set -e
mkdir -p gpurun_out/r05gg
timeout -k 10 600 python -m pytest tests/test_gpu_train.py -x -q -k "narrow or adam_trajectory or spe_wide_cat or ira_wide_att" -s > gpurun_out/r05gg/train_narrow.txt 2>&1 || { tail -n 60 gpurun_out/r05gg/train_narrow.txt; exit 1; }
grep -a "narrow\|passed\|failed\|loss curve" gpurun_out/r05gg/train_narrow.txt | cut -c1-260 | tail -n 40
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train_kernels.py -x -q -k "narrow or rows64" > gpurun_out/r05gg/parity_narrow.txt 2>&1 || { tail -n 40 gpurun_out/r05gg/parity_narrow.txt; exit 1; }
tail -n 3 gpurun_out/r05gg/parity_narrow.txt
