#!/bin/bash
set -e
mkdir -p gpurun_out/r05ll
timeout -k 10 400 python tools/train_energy_ledger.py --seconds 2.5 > gpurun_out/r05ll/train_ledger.txt 2>&1 || { tail -n 30 gpurun_out/r05ll/train_ledger.txt; exit 1; }
tail -n 10 gpurun_out/r05ll/train_ledger.txt
