#!/bin/bash
# Round 4, call O: the 256 x 256 LSTM weight-gradient kernel -- kernel tests, gradient goldens, same-box A/B of the training step.
set -o pipefail
o=gpurun_out/r04o2; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_train_kernels.py -x -q -m gpu > $o/train_kernel_tests.txt 2>&1; echo "train kernel tests rc=$?"; tail -3 $o/train_kernel_tests.txt
timeout -k 10 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu > $o/train_tests.txt 2>&1; echo "train tests rc=$?"; tail -3 $o/train_tests.txt
for i in 1 2 3; do for w in 1 0; do
  r=$(DPRNN_WGRAD_LSTM_WIDE=$w timeout -k 10 300 python bench.py --mode train --steps 6 --warmup 2 --no-power 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],1))")
  echo "round $i wide=$w: $r" | tee -a $o/ab_wgrad_wide.txt
done; done
