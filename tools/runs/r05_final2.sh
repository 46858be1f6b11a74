#!/bin/bash
# round 5, second session: the driver's sequence on the tree with the any-width path: GPU suite timed, smoke, default bench; then the
# any-width path's evidence at F = H = 256 (bench, kernel trace, PMC traffic)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r05f5; mkdir -p $O
( time timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=15 ) > $O/gpu_tests.txt 2>&1 || { tail -40 $O/gpu_tests.txt; exit 1; }
grep -E "passed|failed|real" $O/gpu_tests.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 || { tail $O/smoke.txt; exit 1; }
tail -2 $O/smoke.txt
( time timeout -k 10 900 python bench.py ) > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05f5/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'kernel_ms', d['roofline']['kernel_ms'], 'traffic', d['roofline']['traffic'])
print('exact', d['exact_fp32']['value'], 'cpu', d['cpu_baseline']['value'])
oc=d['other_configs']
print({k:(round(v['value'],1), round(v['ms_per_step'],1)) for k,v in oc.items() if isinstance(v,dict) and 'value' in v}, oc['seconds_spent'])
print('ira roofline', oc['ira_full_length']['roofline']['frac'], 'wide roofline', oc['tasnet_f256_h256_b64']['roofline'])
PY
grep real $O/bench.err
W="tools/bench_wide.py 256 256 64"
timeout -k 10 300 python $W > $O/bench_wide_256.txt 2>&1 || { tail -5 $O/bench_wide_256.txt; exit 1; }
timeout -k 10 300 python tools/bench_wide_rnn.py 256 256 64 >> $O/bench_wide_256.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python3 $W > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
python tools/rocpd_kernel_stats.py $O/trace/t_results.db > $O/wide_kernel_stats.csv || exit 1
head -8 $O/wide_kernel_stats.csv; rm -rf $O/trace
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $W > $O/pmc_fetch.log 2>&1 || { tail -5 $O/pmc_fetch.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $W > $O/pmc_write.log 2>&1 || { tail -5 $O/pmc_write.log; exit 1; }
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 $W (two separate passes, MI355X, DPRNN-TasNet F = H = 256, B = 64 x 3 s, any-width kernels; 6 forwards in the run)" > $O/wide_pmc_traffic.json || exit 1
rm -rf $O/pmc_fetch $O/pmc_write
python -c "import json; d=json.load(open('$O/wide_pmc_traffic.json')); [print(k, v) for k, v in list(d['per_launch'].items())[:6]]"
