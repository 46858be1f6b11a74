for cfg in "16 1" "16 2" "16 4" "8 1" "8 2" "12 1" "12 2" "20 1" "20 2" "24 2" "24 3" "32 2" "32 4" "48 3" "48 4"; do
  set -- $cfg
  r=$(timeout -k 10 120 python bench.py --batch $1 --steps 8 --warmup 3 --no-cpu-baseline --no-optional-modes --groups $2 --min-group 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],1))")
  echo "B=$1 groups=$2 : $r"
done
