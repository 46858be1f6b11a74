#!/bin/bash
# round 5: kernel statistics + timeline of the whole-share ragged IRA run (375 utterances, one ragged batch): slab launches against the rest
set -o pipefail
O=gpurun_out/r05ww; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
FL_ONLY=share timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python3 tools/bench_fulllength.py ira > $O/trace.log 2>&1 || { tail -n 20 $O/trace.log; exit 1; }
grep -a "whole share" $O/trace.log
python tools/rocpd_kernel_stats.py $O/trace/t_results.db --timeline $O/timeline.csv > $O/kernel_stats.csv; head -n 22 $O/kernel_stats.csv | cut -c1-150; rm -rf $O/trace
wc -l $O/timeline.csv
