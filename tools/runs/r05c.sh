#!/bin/bash
# round 5, call C: launch structures of the direction-serial form, same box, interleaved twice
set -o pipefail
O=gpurun_out/r05c; mkdir -p $O
B="--steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-optional-modes --no-power"
run() {  # name, env..., -- args
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py $B $EXTRA > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  python -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],2))"
}
for i in 1 2; do
  EXTRA="" run conc_$i DPRNN_SERIAL_DIRS=0 || exit 1
  EXTRA="" run ser2g_$i DPRNN_SERIAL_DIRS=1 || exit 1
  EXTRA="--groups 1" run ser1g_$i DPRNN_SERIAL_DIRS=1 || exit 1
  EXTRA="--serial-tile-intra 2048" run ser2g_i48_$i DPRNN_SERIAL_DIRS=1 || exit 1
  EXTRA="--serial-tile-intra 2048" run ser2g_i48_stag_$i DPRNN_SERIAL_DIRS=1 DPRNN_SERIAL_STAGGER=1 || exit 1
  EXTRA="" run ser2g_stag_$i DPRNN_SERIAL_DIRS=1 DPRNN_SERIAL_STAGGER=1 || exit 1
  EXTRA="--groups 4 --min-group 16" run ser4g_$i DPRNN_SERIAL_DIRS=1 DPRNN_SERIAL_MIN_FILL=0.1 || exit 1
done
