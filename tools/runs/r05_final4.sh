#!/bin/bash
# round 5, third session, final tree: GPU suite timed, smoke, default bench, kernel trace of the same bench command (eager launches)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/${1:-r05f7}; mkdir -p $O
( time timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=12 ) > $O/gpu_tests.txt 2>&1 || { tail -40 $O/gpu_tests.txt; exit 1; }
grep -E "passed|failed|real" $O/gpu_tests.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 || { tail $O/smoke.txt; exit 1; }
tail -2 $O/smoke.txt
( time timeout -k 10 900 python bench.py ) > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - $O/bench.json <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'kernel_ms', d['roofline']['kernel_ms'], 'traffic', d['roofline']['traffic'])
sa=d['roofline'].get('standalone') or {}
print('standalone', {k: (round(v, 4) if isinstance(v, float) else v) for k, v in sa.items() if k != 'note'})
print('exact', d['exact_fp32']['value'], 'cpu', d['cpu_baseline']['value'])
oc=d['other_configs']
print({k:(round(v['value'],1), round(v['ms_per_step'],1)) for k,v in oc.items() if isinstance(v,dict) and 'value' in v}, oc['seconds_spent'])
PY
grep real $O/bench.err
export DPRNN_GRAPHS=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-optional-modes --no-power --no-standalone > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
python tools/rocpd_kernel_stats.py $O/trace/t_results.db > $O/bench_kernel_stats.csv || exit 1
head -6 $O/bench_kernel_stats.csv | cut -c1-200; rm -rf $O/trace
