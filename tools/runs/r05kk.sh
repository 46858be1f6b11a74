#!/bin/bash
set -e
mkdir -p gpurun_out/r05kk
timeout -k 10 300 python tools/bench_spe.py cat 64 > gpurun_out/r05kk/spe.txt 2>&1 || { tail -n 30 gpurun_out/r05kk/spe.txt; exit 1; }
cat gpurun_out/r05kk/spe.txt | grep -v amdgpu.ids
timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-configs --no-optional-modes --no-power 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bss', round(d['value'],1), round(d['ms_per_step'],2))"
