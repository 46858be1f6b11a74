#!/bin/bash
# mode-1 bench sweep: groups x tiles
out=gpurun_out/r02d_split_sweep.txt; : > $out
run() { echo "== $*" >> $out; timeout -k 10 150 python bench.py --gemm-mode 1 --steps 4 --warmup 2 --no-cpu-baseline --no-optional-modes "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1),'utt/s', round(d['ms_per_step'],1),'ms', 'kern_ms', round(d['roofline']['kernel_ms'],2), 'conc', d['roofline']['mean_concurrency'])" >> $out 2>&1 || return 1; }
run || exit 1
run --seq-tile-intra 64 --seq-tile-inter 64 || exit 1
run --seq-tile-intra 80 --seq-tile-inter 64 || exit 1
run --groups 2 --seq-tile-intra 64 --seq-tile-inter 64 || exit 1
run --groups 2 --seq-tile-intra 80 --seq-tile-inter 64 || exit 1
run --groups 2 --seq-tile-intra 32 --seq-tile-inter 32 || exit 1
run --groups 4 --seq-tile-intra 32 --seq-tile-inter 32 || exit 1
run --groups 4 --seq-tile-intra 64 --seq-tile-inter 64 || exit 1
run --groups 3 --seq-tile-intra 64 --seq-tile-inter 64 || exit 1
run --groups 2 --seq-tile-intra -48 --seq-tile-inter -48 || exit 1
run --groups 4 --seq-tile-intra -48 --seq-tile-inter -48 || exit 1
cat $out
