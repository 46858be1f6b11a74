#!/bin/bash
# round 5, call F: kernel trace + HBM counters of the time-split fold (and the unfolded form's counters on the same box)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r05f; mkdir -p $O
export DPRNN_GRAPHS=0
B="--steps 3 --warmup 1 --no-cpu-baseline --no-optional-modes --no-other-configs"
for f in split off; do
  DPRNN_LIN_FOLD=$f timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace_$f -o t -- python3 bench.py $B > $O/trace_$f.log 2>&1 || { tail -5 $O/trace_$f.log; exit 1; }
  python tools/rocpd_kernel_stats.py $O/trace_$f/t_results.db --timeline $O/timeline_$f.csv > $O/kernel_stats_$f.csv || exit 1
  head -8 $O/kernel_stats_$f.csv
  rm -rf $O/trace_$f
done
P="--steps 1 --warmup 1 --no-cpu-baseline --no-optional-modes --no-other-configs"
for f in split off; do
  DPRNN_LIN_FOLD=$f timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$f -- python3 bench.py $P > $O/pmc_fetch_$f.log 2>&1 || { tail -5 $O/pmc_fetch_$f.log; exit 1; }
  DPRNN_LIN_FOLD=$f timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$f -- python3 bench.py $P > $O/pmc_write_$f.log 2>&1 || { tail -5 $O/pmc_write_$f.log; exit 1; }
  python tools/pmc_traffic.py $O/pmc_fetch_$f $O/pmc_write_$f "DPRNN_LIN_FOLD=$f rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py $P (two separate passes, MI355X, B=64, gate-GEMM mode 3, two utterance groups of 32, eager launches)" bilstm_split8_kernel_f16 32 1552000 2 12 > $O/pmc_traffic_$f.json || exit 1
  rm -rf $O/pmc_fetch_$f $O/pmc_write_$f
  python -c "import json; d=json.load(open('$O/pmc_traffic_$f.json')); print('$f', d['forward'])"
done
