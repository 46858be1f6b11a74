#!/bin/bash
# A/B two builds with bench.py only, interleaved rounds on ONE box.  usage: tools/ab_bench.sh libA.so libB.so [rounds] [extra bench args]
A=$1; B=$2; N=${3:-3}; shift 3
for i in $(seq $N); do
  for L in $A $B; do
    b=$(DPRNN_LIB=$PWD/$L timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],1))")
    echo "round $i $(basename $L): $b"
  done
done
