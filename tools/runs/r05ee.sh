#!/bin/bash
# round 5: the forward's energy ledger before (unfolded) and after (Linear folded in) on one box
set -e
mkdir -p gpurun_out/r05ee
timeout -k 10 280 python tools/forward_energy_ledger.py --fold off  > gpurun_out/r05ee/ledger_off.txt 2> gpurun_out/r05ee/off.err
timeout -k 10 280 python tools/forward_energy_ledger.py --fold auto > gpurun_out/r05ee/ledger_auto.txt 2> gpurun_out/r05ee/auto.err
tail -n 12 gpurun_out/r05ee/ledger_off.txt; tail -n 16 gpurun_out/r05ee/ledger_auto.txt
