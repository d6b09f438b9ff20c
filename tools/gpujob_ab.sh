for i in 1 2 3; do
  echo -n "glds: "; python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['roofline']['mean_us'],1))"
  echo -n "legacy: "; RMEM_PV_LEGACY=1 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['roofline']['mean_us'],1))"
done
