#!/bin/bash
# round 5, call MM: the driver sequence on the tree with narrow-model training: GPU suite timed, smoke, default bench; trace + PMC evidence
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r05mm; mkdir -p $O
( time timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=15 ) > $O/gpu_tests.txt 2>&1 || { tail -40 $O/gpu_tests.txt; exit 1; }
grep -E "passed|failed|real" $O/gpu_tests.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 || { tail $O/smoke.txt; exit 1; }
tail -2 $O/smoke.txt
( time timeout -k 10 900 python bench.py ) > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05mm/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'kernel_ms', d['roofline']['kernel_ms'], 'traffic', d['roofline']['traffic'])
print('exact', d['exact_fp32']['value'], 'cpu', d['cpu_baseline']['value'])
oc=d['other_configs']
print({k:(round(v['value'],1), round(v['ms_per_step'],1)) for k,v in oc.items() if isinstance(v,dict) and 'value' in v}, oc['seconds_spent'])
PY
grep real $O/bench.err
export DPRNN_GRAPHS=0
B="--steps 3 --warmup 1 --no-cpu-baseline --no-optional-modes --no-other-configs --no-power"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python3 bench.py $B > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
python tools/rocpd_kernel_stats.py $O/trace/t_results.db --timeline $O/timeline.csv > $O/kernel_stats.csv || exit 1
head -6 $O/kernel_stats.csv; rm -rf $O/trace
P="--steps 1 --warmup 1 --no-cpu-baseline --no-optional-modes --no-other-configs --no-power"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py $P > $O/pmc_fetch.log 2>&1 || { tail -5 $O/pmc_fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py $P > $O/pmc_write.log 2>&1 || { tail -5 $O/pmc_write.log; exit 1; }
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py $P (two separate passes, MI355X, B=64, gate-GEMM mode 3, Linear folded in (time-split), two utterance groups of 32, eager launches)" bilstm_split8_kernel_f16 32 1552000 2 12 > $O/pmc_traffic.json || exit 1
rm -rf $O/pmc_fetch $O/pmc_write
python -c "import json; d=json.load(open('$O/pmc_traffic.json')); print(d['forward']); print(d['bilstm_split8_kernel_f16'])"
