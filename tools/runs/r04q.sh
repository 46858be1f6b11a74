#!/bin/bash
# Round 4, call Q: wgrad_f16 without MFMAs on padded waves -- train kernel tests, gradient goldens, the training line.
set -o pipefail
o=gpurun_out/r04q2; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_train_kernels.py tests/test_gpu_train.py -x -q -m gpu > $o/train_tests.txt 2>&1; echo "train tests rc=$?"; tail -3 $o/train_tests.txt
for i in 1 2; do
  timeout -k 10 400 python bench.py --mode train --no-power > $o/train_$i.json 2> $o/train.err; python -c "import json; d=json.load(open('$o/train_$i.json')); print(d['value'], d['ms_per_step'])"
done
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r04q2_trace_train -o t -- python3 bench.py --mode train --steps 3 --warmup 1 --no-power > $o/trace_train.log 2>&1
db=$(ls gpurun_out/r04q2_trace_train/*_results.db | head -1); python tools/rocpd_kernel_stats.py $db > $o/train_kernel_stats.csv; head -9 $o/train_kernel_stats.csv | cut -c1-120; rm -rf gpurun_out/r04q2_trace_train
