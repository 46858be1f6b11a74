#!/bin/bash
# Round 4, call A: parity of the shared-reciprocal cell update, energy budget of the recurrence step, same-box A/B against the separate forms.
set -o pipefail
o=gpurun_out/r04a; mkdir -p $o
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q -m gpu > $o/tests.txt 2>&1 || { tail -30 $o/tests.txt; exit 1; }
tail -2 $o/tests.txt
: > $o/energy.jsonl
for v in full:default no_weight_stream:diag1 quarter_lds_reads:diag2 mfma_1_of_24:diag4 no_cell_arithmetic:diag8 no_x_staging_h_stores:diag16 all_removed:diag31 x_separate_reciprocals:sep; do
  label=${v%%:*}; lib=${v#*:}
  if [ $lib = default ]; then unset DPRNN_LIB; else export DPRNN_LIB=$PWD/tools/probes/diag/libdprnn_$lib.so; fi
  timeout -k 10 120 python tools/energy_budget.py --label $label --seconds 3 2>>$o/energy.err | tail -1 >> $o/energy.jsonl || exit 1
  echo "done $label"
done
unset DPRNN_LIB
python tools/energy_budget.py --summarise $o/energy.jsonl | tee $o/energy_budget.txt
tools/ab_bench.sh tss-with-dprnn_amd/libdprnn_mi355x.so tools/probes/diag/libdprnn_sep.so 3 > $o/ab_shared_vs_sep.txt 2>&1
cat $o/ab_shared_vs_sep.txt
