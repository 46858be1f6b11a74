#!/bin/bash
# bench.py in the bf16-split mode (engine default) over utterance-group counts and recurrence tiles
out=gpurun_out/${1:-r02k}_split_sweep.txt; : > $out
run() { echo "== $*" >> $out; timeout -k 10 150 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-optional-modes "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1),'utt/s', round(d['ms_per_step'],1),'ms', 'kern_ms', round(d['roofline']['kernel_ms'],2), 'conc', d['roofline']['mean_concurrency'], 'frac', round(d['roofline']['frac'],3))" >> $out 2>&1 || return 1; }
run || exit 1
run --groups 1 || exit 1
run --groups 2 || exit 1
run --groups 3 || exit 1
run --groups 4 || exit 1
run --groups 4 --seq-tile-intra 2064 --seq-tile-inter 2064 || exit 1
run --groups 3 --seq-tile-intra 2064 --seq-tile-inter 2064 || exit 1
run --groups 2 --seq-tile-intra 2080 --seq-tile-inter 2064 || exit 1
run --gemm-mode 0 || exit 1
cat $out
