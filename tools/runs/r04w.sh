#!/bin/bash
# Round 4, call W: joule ledger of one forward by kernel class.
set -o pipefail
o=gpurun_out/r04w; mkdir -p $o
timeout -k 10 400 python tools/forward_energy_ledger.py --seconds 3 > $o/forward_energy_ledger.txt 2>&1; echo "rc=$?"; cat $o/forward_energy_ledger.txt | cut -c1-220
