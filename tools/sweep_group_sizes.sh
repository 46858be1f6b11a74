for gs in "16,16,16,16" "20,18,14,12" "24,16,14,10" "22,18,14,10" "21,21,22" "18,16,16,14" "16,16,16,16"; do
  r=$(timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-optional-modes --group-sizes $gs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],1), round(d['roofline']['frac'],3), round(d['roofline']['mean_concurrency'],2))")
  echo "sizes=$gs : $r"
done
