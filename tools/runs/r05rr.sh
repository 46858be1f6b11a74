#!/bin/bash
set -o pipefail
O=gpurun_out/r05rr; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_train.py -x -q -k "f16_pieces or exact_fp32" > $O/train_tests.txt 2>&1 || { tail -n 40 $O/train_tests.txt; exit 1; }
tail -n 2 $O/train_tests.txt
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --mode train --no-power 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train', round(d['ms_per_step'],1), 'ms')" | tee -a $O/ab.txt
done
