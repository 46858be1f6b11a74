#!/bin/bash
# round 5: kernel statistics of the training step on the final tree (rocprofv3 --kernel-trace --stats), and its bench line on the same box
set -o pipefail
o=gpurun_out/r05ii; mkdir -p $o
timeout -k 10 400 python bench.py --mode train --no-power > $o/train.json 2> $o/train.err || { tail -n 20 $o/train.err; exit 1; }
python -c "import json; d=json.load(open('$o/train.json')); print('train', d['value'], d['ms_per_step'])"
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r05ii_trace_train -o t -- python3 bench.py --mode train --steps 3 --warmup 1 --no-power > $o/trace_train.log 2>&1 || { tail -n 20 $o/trace_train.log; exit 1; }
db=$(ls gpurun_out/r05ii_trace_train/*_results.db | head -1); python tools/rocpd_kernel_stats.py $db > $o/train_kernel_stats.csv; head -n 14 $o/train_kernel_stats.csv | cut -c1-140; rm -rf gpurun_out/r05ii_trace_train
