#!/bin/bash
# round 5, call K: non-temporal cache policy on the folded launches' activation streams (A/B on one box, bench + stand-alone phases)
set -o pipefail
O=gpurun_out/r05k; mkdir -p $O
B="--steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-optional-modes --no-power"
for i in 1 2; do
  for v in base nt1 nt6 nt9 nt15; do
    DPRNN_LIB=$PWD/tools/probes/diag/libdprnn_lin_$v.so timeout -k 10 300 python bench.py $B > $O/b_${v}_$i.json 2> $O/b_${v}_$i.err || { tail -5 $O/b_${v}_$i.err; exit 1; }
    python -c "import json; d=json.loads(open('$O/b_${v}_$i.json').read().strip().splitlines()[-1]); print('$v $i', round(d['value'],1), round(d['ms_per_step'],2), round(d['roofline']['frac'],4), round(d['roofline']['kernel_ms'],3))"
  done
done
for v in base nt15; do
  echo "== $v"; DPRNN_LIB=$PWD/tools/probes/diag/libdprnn_lin_$v.so timeout -k 10 200 python tools/bench_serial.py --batch 32 2>&1 | grep "time-split"
done
