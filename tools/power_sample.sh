#!/bin/bash
# Samples the GPU's power draw and clocks (rocm-smi) while a command runs: tools/power_sample.sh OUT -- cmd args...
out=$1; shift; shift
"$@" > ${out}.cmd.txt 2>&1 &
pid=$!
for i in $(seq 1 400); do
    rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|edge)" | tr '\n' ' ' ; echo
    sleep 0.5
    kill -0 $pid 2>/dev/null || break
done > ${out}.smi.txt
wait $pid
