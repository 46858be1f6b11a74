#!/bin/bash
# round 5, call Y: cache policy of the BPTT kernel's once-touched streams (non-temporal loads / stores), training step A/B on one box
set -o pipefail
for i in 1 2; do for v in base st0 ld0st0; do
  DPRNN_LIB=$PWD/tools/probes/diag/libdprnn_bwd_$v.so timeout -k 10 300 python bench.py --mode train --steps 6 --warmup 3 --no-power 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bwd $v', round(d['value'],1), round(d['ms_per_step'],1), round(d['roofline']['kernel_ms'],3))"
done; done
