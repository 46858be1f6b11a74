#!/bin/bash
# Round 4, call P: rocprofv3 kernel trace of the training step (wide weight-gradient kernel) + the bench's own line on the same box.
set -o pipefail
o=gpurun_out/r04p2; mkdir -p $o
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r04p2_trace_train -o t -- python3 bench.py --mode train --steps 3 --warmup 1 --no-power > $o/trace_train.log 2>&1 || { tail -5 $o/trace_train.log; exit 1; }
db=$(ls gpurun_out/r04p2_trace_train/*_results.db | head -1)
python tools/rocpd_kernel_stats.py $db > $o/train_kernel_stats.csv; head -12 $o/train_kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/r04p2_trace_train
timeout -k 10 400 python bench.py --mode train > $o/train.json 2> $o/train.err; python -c "import json; d=json.load(open('$o/train.json')); print(d['value'], d['ms_per_step'], d['power'])"
DPRNN_WGRAD_LSTM_WIDE=0 timeout -k 10 400 python bench.py --mode train > $o/train_narrow.json 2> $o/train.err; python -c "import json; d=json.load(open('$o/train_narrow.json')); print(d['value'], d['ms_per_step'])"
