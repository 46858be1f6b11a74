#!/bin/bash
# round 5: fused GroupNorm-pass-2 + Linear backward (dprnn_lin_bwd_f16): kernel test, gradient goldens, the training step with and without
set -o pipefail
O=gpurun_out/r05pp; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_train_kernels.py -x -q -k "fused_norm_and_linear" -s > $O/kernel_test.txt 2>&1 || { tail -n 40 $O/kernel_test.txt; exit 1; }
grep -a "rows/utt\|passed" $O/kernel_test.txt
timeout -k 10 600 python -m pytest tests/test_gpu_train.py -x -q -k "f16_pieces and (training_step_matches or spe_training_step or b64_full)" > $O/train_tests.txt 2>&1 || { tail -n 40 $O/train_tests.txt; exit 1; }
tail -n 2 $O/train_tests.txt
for i in 1 2; do
  timeout -k 10 300 python bench.py --mode train --no-power 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused', round(d['ms_per_step'],1), 'ms')" | tee -a $O/ab.txt
  DPRNN_TRAIN_LIN_BWD_FUSED=0 timeout -k 10 300 python bench.py --mode train --no-power 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('three launches', round(d['ms_per_step'],1), 'ms')" | tee -a $O/ab.txt
done
