#!/bin/bash
# Round 4, call X: forward time at B = 17..32 with the inter-chunk tile forced (16- against 64-sequence tiles), mode 3.
set -o pipefail
o=gpurun_out/r04x; mkdir -p $o
timeout -k 10 500 python tools/sweep_inter_tile.py 17:34:1 0 0,2016,2064 > $o/inter_tile_sweep.txt 2>&1; echo "rc=$?"; grep "^B=" $o/inter_tile_sweep.txt | cut -c1-200
timeout -k 10 300 python tools/sweep_inter_tile.py 24:40:2 2016,2064 2064 > $o/intra_tile_sweep.txt 2>&1; echo "rc=$?"; grep "^B=" $o/intra_tile_sweep.txt | cut -c1-200
