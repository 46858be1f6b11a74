#!/bin/bash
set -o pipefail
O=gpurun_out/r05qq; mkdir -p $O
( time timeout -k 10 900 python -m pytest tests/test_gpu_train_kernels.py tests/test_gpu_train.py -x -q --durations=8 ) > $O/train_tests.txt 2>&1 || { tail -n 40 $O/train_tests.txt; exit 1; }
tail -n 14 $O/train_tests.txt
