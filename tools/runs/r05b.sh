#!/bin/bash
# round 5, call B: where the direction-serial form loses its time -- one path alone in both structures, then kernel traces of both
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r05b; mkdir -p $O
timeout -k 10 300 python tools/bench_serial.py --batch 32,64 > $O/bench_serial.txt 2>&1 || { tail -20 $O/bench_serial.txt; exit 1; }
cat $O/bench_serial.txt
B="--steps 3 --warmup 1 --no-cpu-baseline --no-optional-modes --no-other-configs"
for s in 1 0; do
  DPRNN_GRAPHS=0 DPRNN_SERIAL_DIRS=$s timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace_s$s -o t -- python3 bench.py $B > $O/trace_s$s.log 2>&1 || { tail -5 $O/trace_s$s.log; exit 1; }
  python tools/rocpd_kernel_stats.py $O/trace_s$s/*/t_results.db --timeline $O/timeline_s$s.csv > $O/kernel_stats_s$s.csv || exit 1
  head -12 $O/kernel_stats_s$s.csv
  rm -rf $O/trace_s$s
done
