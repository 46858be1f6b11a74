#!/bin/bash
set -o pipefail
O=gpurun_out/r05u; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_spe.py -x -q -m gpu -k "headline or tasnet_matches or graph or narrow_twin or spe_matches or ira or stream_groups" > $O/parity.txt 2>&1 || { tail -30 $O/parity.txt; exit 1; }
tail -2 $O/parity.txt
Bn="--steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-optional-modes --no-power"
for i in 1 2 3; do for p in 0 1; do
  DPRNN_PREPACK=$p python bench.py $Bn 2> $O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('prepack=$p', round(d['value'],1), round(d['ms_per_step'],3))" || { tail -3 $O/err.txt; exit 1; }
done; done
