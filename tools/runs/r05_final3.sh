#!/bin/bash
# round 5, third session: the driver's sequence on the final tree (GPU suite timed, the `slow` repetitions once, smoke, default bench
# with the live stand-alone launch timing in its roofline object), then the kernel trace of the same bench command
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r05f6; mkdir -p $O
( time timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=15 ) > $O/gpu_tests.txt 2>&1 || { tail -40 $O/gpu_tests.txt; exit 1; }
grep -E "passed|failed|real" $O/gpu_tests.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 || { tail $O/smoke.txt; exit 1; }
tail -2 $O/smoke.txt
( time timeout -k 10 900 python bench.py ) > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05f6/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'kernel_ms', d['roofline']['kernel_ms'], 'traffic', d['roofline']['traffic'])
print('standalone', d['roofline'].get('standalone'))
print('exact', d['exact_fp32']['value'], 'cpu', d['cpu_baseline']['value'])
oc=d['other_configs']
print({k:(round(v['value'],1), round(v['ms_per_step'],1)) for k,v in oc.items() if isinstance(v,dict) and 'value' in v}, oc['seconds_spent'])
PY
grep real $O/bench.err
( time timeout -k 10 400 python -m pytest tests -x -q -m "gpu and slow" ) > $O/gpu_tests_slow.txt 2>&1 || { tail -40 $O/gpu_tests_slow.txt; exit 1; }
grep -E "passed|failed|real" $O/gpu_tests_slow.txt
