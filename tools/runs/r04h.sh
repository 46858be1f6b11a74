#!/bin/bash
# Round 4, call H: parity / spe / inference tests with the narrow-width twins, then the rocprofv3 evidence of the bench line
# (kernel trace + stats in the three modes, PMC traffic of the forward) and the bench on the same box.
set -o pipefail
o=gpurun_out/r04h; mkdir -p $o
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_spe.py tests/test_gpu_inference.py -x -q -m gpu > $o/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $o/tests.txt
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
bash tools/collect_profiles.sh r04p > $o/collect.log 2>&1; tail -2 $o/collect.log
for m in 3 1 0; do
  db=$(ls gpurun_out/r04p_trace_mode$m/*_results.db 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/rocpd_kernel_stats.py $db $([ $m = 3 ] && echo --timeline $o/mode3_timeline.csv) > $o/mode${m}_kernel_stats.csv
done
python tools/timeline_summary.py $o/mode3_timeline.csv > $o/timeline_summary.txt 2>&1
cp gpurun_out/r04p_pmc_traffic.json $o/pmc_traffic_f16.json 2>/dev/null
rm -rf gpurun_out/r04p_trace_mode*
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $o/bench_same_box.json 2> $o/bench.err; python -c "import json; d=json.load(open('$o/bench_same_box.json')); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['exact_fp32']['value'])"
head -4 $o/mode3_kernel_stats.csv
