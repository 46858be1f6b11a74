#!/bin/bash
# HBM traffic of the training step's kernels (rocprofv3 --pmc, two separate passes) -> gpurun_out/$1_pmc_traffic_train.json
tag=${1:-r04}
export TMPDIR=/tmp
set -o pipefail
P="--mode train --steps 1 --warmup 1 --no-cpu-baseline"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_tfetch -- python3 bench.py $P > gpurun_out/${tag}_pmc_tfetch.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_twrite -- python3 bench.py $P > gpurun_out/${tag}_pmc_twrite.log 2>&1 || exit 1
python tools/pmc_traffic.py gpurun_out/${tag}_pmc_tfetch gpurun_out/${tag}_pmc_twrite "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py $P (two separate passes, MI355X, 64 utterances per step as two micro-batches of 32; forward, BPTT and weight gradients in f16 pieces)" > gpurun_out/${tag}_pmc_traffic_train.json || exit 1
rm -rf gpurun_out/${tag}_pmc_tfetch gpurun_out/${tag}_pmc_twrite
echo collected
