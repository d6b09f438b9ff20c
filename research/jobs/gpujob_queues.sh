for q in 4 8 16; do for c in 1 2 4; do
  echo -n "queues=$q clips=$c: "
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --clips-per-gpu $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3))"
done; done
