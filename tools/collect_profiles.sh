#!/bin/bash
# Runs on the GPU box (gpurun): the evidence behind bench.py's numbers.  Outputs under gpurun_out/$1_*; tools/rocpd_kernel_stats.py
# and tools/pmc_traffic.py turn them into the files committed under profiles/.
tag=${1:-r04}
export TMPDIR=/tmp
set -o pipefail
export DPRNN_GRAPHS=0        # eager launches: one trace record per kernel of every forward (a replayed graph is one launch)
B="--steps 3 --warmup 1 --no-cpu-baseline --no-optional-modes --no-standalone"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_trace_mode3 -o t -- python3 bench.py $B > gpurun_out/${tag}_trace_mode3.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_trace_mode1 -o t -- python3 bench.py $B --gemm-mode 1 > gpurun_out/${tag}_trace_mode1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_trace_mode0 -o t -- python3 bench.py $B --gemm-mode 0 > gpurun_out/${tag}_trace_mode0.log 2>&1 || exit 1
P="--steps 1 --warmup 1 --no-cpu-baseline --no-optional-modes --no-standalone"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_fetch -- python3 bench.py $P > gpurun_out/${tag}_pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_write -- python3 bench.py $P > gpurun_out/${tag}_pmc_write.log 2>&1 || exit 1
python tools/pmc_traffic.py gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py $P (two separate passes, MI355X, B=64, default gate-GEMM mode 3: f16 pieces, two utterance groups of 32)" bilstm_split8_kernel_f16 32 1552000 > gpurun_out/${tag}_pmc_traffic.json || exit 1
rm -rf gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write
echo collected
