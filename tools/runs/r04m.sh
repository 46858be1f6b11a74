#!/bin/bash
# Round 4, call M: step costs of the exact kernel's tiles, and tile / group variants at the batch sizes where mode 0 still dips.
set -o pipefail
o=gpurun_out/r04m2; mkdir -p $o
timeout -k 10 200 python tools/bench_recur.py > $o/exact_tiles.txt 2>&1; cat $o/exact_tiles.txt
run() { r=$(timeout -k 10 120 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-optional-modes --no-power --gemm-mode 0 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), d['roofline']['concurrent_streams'])"); echo "mode 0 $*: $r" | tee -a $o/variants.txt; }
for B in 8 9 10 14; do
  run --batch $B
  run --batch $B --seq-tile-inter 32
  run --batch $B --seq-tile-intra 32 --seq-tile-inter 32
  run --batch $B --groups 2 --min-group 2 --seq-tile-inter 32
done
