#!/bin/bash
# Round 4, call C: telescoping energy budget (components removed cumulatively: closes by construction), graph tests, bench with
# other_configs, exact-mode B=64 alone, graph churn walk with the shared pool, then the legacy arm once (expected to crash).
set -o pipefail
o=gpurun_out/r04c; mkdir -p $o
: > $o/energy_cumulative.jsonl
for v in full:default minus_h_store:diag32 minus_x_staging:diag48 minus_cell_arithmetic:diag56 minus_weight_stream:diag57 minus_3_of_4_lds_reads:diag59 minus_23_of_24_mfma:diag63; do
  label=${v%%:*}; lib=${v#*:}
  if [ $lib = default ]; then unset DPRNN_LIB; else export DPRNN_LIB=$PWD/tools/probes/diag/libdprnn_$lib.so; fi
  timeout -k 10 120 python tools/energy_budget.py --label $label --seconds 2.5 2>>$o/energy.err | tail -1 >> $o/energy_cumulative.jsonl || exit 1
done
unset DPRNN_LIB
python tools/energy_budget.py --cumulative $o/energy_cumulative.jsonl | tee $o/energy_cumulative.txt
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "graph or pool or capture or inference_mode" > $o/graph_tests.txt 2>&1 || { tail -40 $o/graph_tests.txt; exit 1; }
tail -2 $o/graph_tests.txt
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err || { tail -20 $o/bench.err; exit 1; }
python -c "import json; d=json.load(open('$o/bench.json')); print(d['value'], d['exact_fp32']['value'], d['host'].get('reserved_GiB'), json.dumps(d['other_configs'])[:1800])"
timeout -k 10 200 python tools/latency_modes.py 64 > $o/latency_b64_alone.txt 2>&1; cat $o/latency_b64_alone.txt
timeout -k 10 500 python tools/probes/graph_churn.py 4 16,24,32,48,64,96,128 > $o/churn_shared_pool.txt 2>&1; tail -4 $o/churn_shared_pool.txt
CHURN_LEGACY=1 timeout -k 10 400 python tools/probes/graph_churn.py 4 16,24,32,48,64,96,128 > $o/churn_legacy.txt 2>&1; tail -25 $o/churn_legacy.txt | cut -c1-250
true
