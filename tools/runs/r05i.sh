#!/bin/bash
set -o pipefail
O=gpurun_out/r05i; mkdir -p $O
echo "== K=250" >> $O/t.txt
timeout -k 10 200 python tools/bench_serial.py --batch 32,64 2>&1 | grep "B=" >> $O/t.txt || exit 1
echo "== K=1000 (T = 1000 intra: fixed per-launch overhead amortised)" >> $O/t.txt
timeout -k 10 200 python tools/bench_serial.py --batch 32 --K 1000 2>&1 | grep "B=32 intra" >> $O/t.txt || exit 1
echo "== alias P = Z (K=250)" >> $O/t.txt
DPRNN_PROBE_ALIAS=1 timeout -k 10 200 python tools/bench_serial.py --batch 32 2>&1 | grep "time-split" >> $O/t.txt || exit 1
cat $O/t.txt
