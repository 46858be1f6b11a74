#!/bin/bash
# round 5, call M: the full GPU suite under the new defaults (timed), smoke, default bench line
set -o pipefail
O=gpurun_out/r05m; mkdir -p $O
( time timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=25 ) > $O/gpu_tests.txt 2>&1 || { tail -40 $O/gpu_tests.txt; exit 1; }
tail -40 $O/gpu_tests.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 || { tail $O/smoke.txt; exit 1; }
tail -2 $O/smoke.txt
