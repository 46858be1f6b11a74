#!/bin/bash
# round 5: kernel statistics of the full-length DPRNN-Spe-IRA ragged batch (48 utterances): what is outside the slab launches?
set -o pipefail
O=gpurun_out/r05tt; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
FL_ONLY=ragged timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python3 tools/bench_fulllength.py ira > $O/trace.log 2>&1 || { tail -n 20 $O/trace.log; exit 1; }
grep -a "ragged batch" $O/trace.log
python tools/rocpd_kernel_stats.py $O/trace/t_results.db > $O/kernel_stats.csv; head -n 25 $O/kernel_stats.csv | cut -c1-150; rm -rf $O/trace
