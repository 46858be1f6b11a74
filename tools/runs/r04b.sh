#!/bin/bash
# Round 4, call B: finer energy probes (x staging split up), graph pool tests, bench with other_configs, exact-mode B=64 discrepancy.
set -o pipefail
o=gpurun_out/r04b; mkdir -p $o
: > $o/energy.jsonl
for v in full:default no_weight_stream:diag1 mfma_1_of_24:diag4 no_cell_arithmetic:diag8 no_x_staging:diag16 no_h_store:diag32 no_xprime_store:diag64 x_loads_hit_L2:diag128 no_x_staging_no_h_store:diag48 all_removed:diag63; do
  label=${v%%:*}; lib=${v#*:}
  if [ $lib = default ]; then unset DPRNN_LIB; else export DPRNN_LIB=$PWD/tools/probes/diag/libdprnn_$lib.so; fi
  timeout -k 10 120 python tools/energy_budget.py --label $label --seconds 2.5 2>>$o/energy.err | tail -1 >> $o/energy.jsonl || exit 1
done
: > $o/energy_nofuse.jsonl
for v in full:default no_x_staging:diag16; do
  label=${v%%:*}; lib=${v#*:}
  if [ $lib = default ]; then unset DPRNN_LIB; else export DPRNN_LIB=$PWD/tools/probes/diag/libdprnn_$lib.so; fi
  timeout -k 10 120 python tools/energy_budget.py --label $label --seconds 2.5 --fuse 0 2>>$o/energy.err | tail -1 >> $o/energy_nofuse.jsonl || exit 1
done
unset DPRNN_LIB
python tools/energy_budget.py --summarise $o/energy.jsonl | tee $o/energy_budget.txt
python tools/energy_budget.py --summarise $o/energy_nofuse.jsonl | tee $o/energy_budget_nofuse.txt
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "graph or pool or capture or inference_mode" > $o/graph_tests.txt 2>&1 || { tail -40 $o/graph_tests.txt; exit 1; }
tail -2 $o/graph_tests.txt
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err || { tail -20 $o/bench.err; exit 1; }
python -c "import json; d=json.load(open('$o/bench.json')); print(d['value'], d['exact_fp32']['value'], d['host'].get('reserved_GiB'), json.dumps(d['other_configs'])[:1500])"
timeout -k 10 200 python tools/latency_modes.py 64 > $o/latency_b64_alone.txt 2>&1; cat $o/latency_b64_alone.txt
