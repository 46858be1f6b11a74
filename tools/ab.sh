#!/bin/bash
# A/B two builds of libdprnn_mi355x.so on ONE GPU box (devices differ by up to ~10 %: never compare across boxes).
# usage: tools/ab.sh libA.so libB.so [rounds]
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for L in $A $B; do
    r=$(DPRNN_LIB=$PWD/$L timeout -k 10 200 python tools/bench_recur.py --only 32,2 2>&1 | tail -1 | awk '{print $(NF-1)}')
    b=$(DPRNN_LIB=$PWD/$L timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")
    echo "round $i $L: recur32x2 $r TFLOP/s, bench $b utt/s"
  done
done
