#!/bin/bash
# round 5, call E: the time-split fold -- kernel tests, parity subset under the new default, bench A/B against no fold on the same box
set -o pipefail
O=gpurun_out/r05e; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "time_split or direction_serial" > $O/kernels.txt 2>&1 || { tail -40 $O/kernels.txt; exit 1; }
tail -2 $O/kernels.txt
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "narrow_twin_with_several or headline_batch or tasnet_matches_reference_golden or graph" > $O/parity.txt 2>&1 || { tail -40 $O/parity.txt; exit 1; }
tail -2 $O/parity.txt
B="--steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-optional-modes --no-power"
run() {
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py $B > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  python -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],2), d['parity']['rel_l2_vs_cpu'] if 'parity' in d and d['parity'] else None)"
}
for i in 1 2 3; do
  run off_$i DPRNN_LIN_FOLD=off || exit 1
  run split_$i DPRNN_LIN_FOLD=split || exit 1
done
