#!/bin/bash
# round 5, call N: the driver's sequence on the tree -- GPU suite (timed, all durations), smoke, default bench line
set -o pipefail
O=gpurun_out/r05n; mkdir -p $O
( time timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=0 ) > $O/gpu_tests.txt 2>&1 || { tail -40 $O/gpu_tests.txt; exit 1; }
grep -E "passed|failed|real" $O/gpu_tests.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 || { tail $O/smoke.txt; exit 1; }
tail -2 $O/smoke.txt
( time timeout -k 10 900 python bench.py ) > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05n/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'kernel_ms', d['roofline']['kernel_ms'], 'traffic', d['roofline']['traffic'])
print('exact', d['exact_fp32']['value'], 'cpu', d['cpu_baseline']['value'], d['parity'])
oc=d['other_configs']
print({k:(round(v['value'],1), round(v['ms_per_step'],1)) for k,v in oc.items() if isinstance(v,dict) and 'value' in v}, oc['seconds_spent'])
PY
tail -3 $O/bench.err
