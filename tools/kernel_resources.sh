#!/bin/bash
# Per-kernel register / scratch listing of one HIP source (gfx950), from the compiler's own remarks.
# usage: tools/kernel_resources.sh tss-with-dprnn_amd/csrc/recur_split8_f16.hip [extra flags]   (part of the change procedure: no spills in step loops)
src=$1; shift
dir=$(dirname "$src")
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I"$dir/../../include" -Wno-unused-function "$@" \
    -Rpass-analysis=kernel-resource-usage -c "$src" -o /dev/null 2>&1 |
  awk '/remark: Function Name:/ {name=$(NF-1)} /remark: +TotalSGPRs:/ {sg=$(NF-1)} /remark: +VGPRs:/ {v=$(NF-1)} /remark: +AGPRs:/ {a=$(NF-1)}
       /remark: +ScratchSize/ {s=$(NF-1)} /remark: +VGPRs Spill:/ {sp=$(NF-1)} /remark: +Occupancy/ {o=$(NF-1)}
       /remark: +LDS Size/ {print name, "VGPRs", v, "AGPRs", a, "SGPRs", sg, "scratch", s, "spill", sp, "waves/SIMD", o}' |
  while read -r n rest; do echo "$(echo "$n" | c++filt | sed 's/(recur::RecurArgs)//; s/^void //' | cut -c1-100) $rest"; done
