#!/bin/bash
# round 5, fourth session: the final tree's kernel trace WITHOUT the stand-alone timing launches (bench.py --no-standalone), the two PMC
# passes of the same command, then the default bench line (with roofline.standalone) on the same box
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r05f8; mkdir -p $O
( time timeout -k 10 600 python bench.py ) > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05f8/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'kernel_ms', d['roofline']['kernel_ms'], 'traffic', d['roofline']['traffic'])
sa=d['roofline'].get('standalone') or {}
print('standalone', {k: (round(v, 4) if isinstance(v, float) else v) for k, v in sa.items() if k != 'note'})
oc=d['other_configs']
print({k:(round(v['value'],1), round(v['ms_per_step'],1)) for k,v in oc.items() if isinstance(v,dict) and 'value' in v}, oc['seconds_spent'])
PY
export DPRNN_GRAPHS=0
A="--no-cpu-baseline --no-optional-modes --no-standalone --no-power"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python3 bench.py --steps 3 --warmup 1 $A > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
python tools/rocpd_kernel_stats.py $O/trace/t_results.db > $O/bench_kernel_stats.csv || exit 1
head -5 $O/bench_kernel_stats.csv | cut -c1-200; tail -1 $O/trace.log | cut -c1-400; rm -rf $O/trace
P="--steps 1 --warmup 1 $A"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py $P > $O/pmc_fetch.log 2>&1 || { tail -5 $O/pmc_fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py $P > $O/pmc_write.log 2>&1 || { tail -5 $O/pmc_write.log; exit 1; }
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py $P (two separate passes, MI355X, B=64, default gate-GEMM mode 3, final tree of round 5)" bilstm_split8_kernel_f16 32 1552000 > $O/pmc_traffic.json || exit 1
rm -rf $O/pmc_fetch $O/pmc_write
python -c "import json; d=json.load(open('$O/pmc_traffic.json')); print({k: d[k] for k in list(d)[:6]})" | cut -c1-600
