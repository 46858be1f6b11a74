#!/bin/bash
# Round 4, call E: the whole GPU suite on the tree so far, then utterance-group counts at the batch sizes where throughput dips.
set -o pipefail
o=gpurun_out/r04e; mkdir -p $o
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $o/tests.txt 2>&1 || { tail -60 $o/tests.txt; exit 1; }
tail -3 $o/tests.txt
for mode in 3 0; do
  for B in 9 11 12 17 22 24 36 44; do
    for G in 1 2 3 4; do
      r=$(timeout -k 10 120 python bench.py --batch $B --steps 8 --warmup 3 --no-cpu-baseline --no-optional-modes --no-power --gemm-mode $mode --groups $G --min-group 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), d['roofline']['concurrent_streams'])")
      echo "mode $mode B=$B groups=$G: $r" | tee -a $o/group_sweep.txt
    done
  done
done
