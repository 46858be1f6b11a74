#!/bin/bash
# round 5, third session: two pipelines of unequal size (the shorter one's mask head / decoder then runs under the longer one's last
# recurrence launches instead of after them) against the even split, alternating on one box
set -o pipefail
O=gpurun_out/r05g; mkdir -p $O
for gs in "32,32" "30,34" "34,30" "28,36" "32,32" "30,34" "34,30" "28,36" "24,40"; do
  r=$(timeout -k 10 150 python bench.py --steps 30 --warmup 4 --no-cpu-baseline --no-optional-modes --no-other-configs --no-power --group-sizes $gs 2>$O/err.txt | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), round(d['ms_per_step'],2), 'busy', round(r['busy_ms_per_step'],2), 'conc', round(r['mean_concurrency'],2), 'kernel_ms', round(r['kernel_ms'],3))") || { tail -n 20 $O/err.txt; exit 1; }
  echo "sizes=$gs : $r" | tee -a $O/uneven.txt
done
