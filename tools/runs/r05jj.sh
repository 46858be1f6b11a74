#!/bin/bash
set -e
mkdir -p gpurun_out/r05jj
timeout -k 10 400 python tools/probes/exact_mode_ledger.py > gpurun_out/r05jj/exact_ledger.txt 2>&1 || { tail -n 30 gpurun_out/r05jj/exact_ledger.txt; exit 1; }
cat gpurun_out/r05jj/exact_ledger.txt
