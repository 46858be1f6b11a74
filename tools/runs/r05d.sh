#!/bin/bash
# round 5, call D: timing probe of the "both directions in one launch, each folds its half of the Linear" structure (results wrong by design)
set -o pipefail
O=gpurun_out/r05d; mkdir -p $O
B="--steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-optional-modes --no-power"
run() {
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py $B > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  python -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],2), d.get('saturation'))"
}
for i in 1 2 3; do
  run conc_$i DPRNN_SERIAL_DIRS=0 DPRNN_SATURATION_CHECK=0 || exit 1
  run probe_$i DPRNN_SERIAL_DIRS=1 DPRNN_PROBE_MEET=1 DPRNN_SATURATION_CHECK=0 || exit 1
done
