for cfg in "32 0 0" "32 32 32" "32 32 64" "32 16 32" "48 0 0" "48 32 32" "48 32 64" "24 0 0" "24 32 32" "16 0 0" "16 32 32"; do
  set -- $cfg
  r=$(timeout -k 10 120 python bench.py --batch $1 --steps 8 --warmup 3 --no-cpu-baseline --no-optional-modes --seq-tile-intra $2 --seq-tile-inter $3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],1), d['roofline']['concurrent_streams'])")
  echo "B=$1 intra=$2 inter=$3 : $r"
done
