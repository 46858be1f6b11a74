# usage: bash tools/sweep_groups.sh   -- stream-group / hardware-queue sweep of bench.py on one box
for cfg in "4 4 0 0" "8 4 0 0" "8 5 0 0" "8 6 0 0" "8 8 0 0" "8 8 32 32" "16 8 32 32" "16 16 32 32" "16 6 32 32" "8 4 32 32" "4 4 32 32"; do
  set -- $cfg
  r=$(GPU_MAX_HW_QUEUES=$1 timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-optional-modes --groups $2 --min-group 4 --seq-tile-intra $3 --seq-tile-inter $4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],1), round(d['roofline']['frac'],3))")
  echo "hwq=$1 groups=$2 intra=$3 inter=$4 : $r"
done
