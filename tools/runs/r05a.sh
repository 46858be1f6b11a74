#!/bin/bash
# round 5, call A: the direction-serial kernels against the oracle, the narrow-twin regression, then serial vs concurrent A/B of the bench
set -o pipefail
O=gpurun_out/r05a; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "direction_serial" -s > $O/kernels.txt 2>&1 || { tail -30 $O/kernels.txt; exit 1; }
tail -3 $O/kernels.txt
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "narrow_twin_with_several or headline_batch or tasnet_matches_reference_golden" > $O/parity.txt 2>&1 || { tail -30 $O/parity.txt; exit 1; }
tail -3 $O/parity.txt
for i in 1 2; do
  for s in 0 1; do
    DPRNN_SERIAL_DIRS=$s timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-optional-modes > $O/bench_s${s}_$i.json 2> $O/bench_s${s}_$i.err || { tail -5 $O/bench_s${s}_$i.err; exit 1; }
    python -c "import json,sys; d=json.loads(open('$O/bench_s${s}_$i.json').read().strip().splitlines()[-1]); print('serial=$s run $i', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))"
  done
done
