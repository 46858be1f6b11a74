#!/bin/bash
# Round 4, call I: exact-fp32 kernel with / without the shared-reciprocal cell (same box A/B), kernel tests, memory of a captured forward.
set -o pipefail
o=gpurun_out/r04i; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > $o/kernel_tests.txt 2>&1; echo "kernel tests rc=$?"; tail -2 $o/kernel_tests.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden and not mode1" > $o/golden_tests.txt 2>&1; echo "golden tests rc=$?"; tail -2 $o/golden_tests.txt
tools/ab_bench.sh tss-with-dprnn_amd/libdprnn_mi355x.so tools/probes/diag/libdprnn_exact_sep.so 3 --gemm-mode 0 --no-optional-modes --no-power 2>&1 | tee $o/ab_exact_shared_vs_sep.txt
for B in 8 16; do for L in tss-with-dprnn_amd/libdprnn_mi355x.so tools/probes/diag/libdprnn_exact_sep.so; do
  r=$(DPRNN_LIB=$PWD/$L timeout -k 10 120 python bench.py --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-optional-modes --no-power --gemm-mode 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2))")
  echo "B=$B $(basename $L): $r" | tee -a $o/ab_exact_shared_vs_sep.txt; done; done
timeout -k 10 200 python tools/probes/graph_memory.py 64 > $o/graph_memory.txt 2>&1; head -9 $o/graph_memory.txt | cut -c1-200
