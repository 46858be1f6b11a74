#!/bin/bash
# Round 4, call F: spe / inference / graph tests after the async + glue work, then utterance-group counts at the dipping batch sizes.
set -o pipefail
o=gpurun_out/r04f; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_spe.py tests/test_gpu_inference.py tests/test_gpu_parity.py -x -q -m gpu > $o/tests.txt 2>&1; echo "tests rc=$?"; tail -5 $o/tests.txt
for mode in 3 0; do
  for B in 9 11 12 17 22 24 36 44; do
    for G in 1 2 3 4; do
      r=$(timeout -k 10 120 python bench.py --batch $B --steps 8 --warmup 3 --no-cpu-baseline --no-optional-modes --no-power --gemm-mode $mode --groups $G --min-group 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), d['roofline']['concurrent_streams'])")
      echo "mode $mode B=$B groups=$G: $r" | tee -a $o/group_sweep.txt
    done
  done
done
