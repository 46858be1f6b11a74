#!/bin/bash
# Round 4, call R: packed gate tape (24-bit fixed-point gates, 4 KB instead of 5 KB per row and direction) -- kernel test first,
# then the train tests, the training line with the packed and with the fp32 tape on the same box, kernel stats.
set -o pipefail
o=gpurun_out/r04r; mkdir -p $o
timeout -k 10 300 python -m pytest tests/test_gpu_train_kernels.py -x -q -m gpu -k "packed_gate_tape or backward" -s > $o/packed_test.txt 2>&1; rc=$?; echo "packed test rc=$rc"; grep -E "packed tape|passed|failed|Error" $o/packed_test.txt | tail -8
[ $rc -eq 0 ] || exit 1
for i in 1 2; do
  timeout -k 10 400 python bench.py --mode train --no-power > $o/train_packed_$i.json 2> $o/train.err; python -c "import json; d=json.load(open('$o/train_packed_$i.json')); print('packed', d['value'], d['ms_per_step'])"
  DPRNN_TRAIN_TAPE_PACKED=0 timeout -k 10 400 python bench.py --mode train --no-power > $o/train_fp32tape_$i.json 2> $o/train.err; python -c "import json; d=json.load(open('$o/train_fp32tape_$i.json')); print('fp32 tape', d['value'], d['ms_per_step'])"
done
timeout -k 10 700 python -m pytest tests/test_gpu_train_kernels.py tests/test_gpu_train.py -x -q -m gpu > $o/train_tests.txt 2>&1; echo "train tests rc=$?"; tail -3 $o/train_tests.txt
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r04r_trace_train -o t -- python3 bench.py --mode train --steps 3 --warmup 1 --no-power > $o/trace_train.log 2>&1
db=$(ls gpurun_out/r04r_trace_train/*_results.db | head -1); python tools/rocpd_kernel_stats.py $db > $o/train_kernel_stats.csv; head -9 $o/train_kernel_stats.csv | cut -c1-120; rm -rf gpurun_out/r04r_trace_train
