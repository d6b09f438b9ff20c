# A/B of the long/short branch issue order under hipGraph replay (RMEM_BRANCH_ORDER)
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
for o in short_first long_first serial; do
  echo -n "$o: "; RMEM_BRANCH_ORDER=$o python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3))"
done; done
for o in long_first serial; do
RMEM_BRANCH_ORDER=$o rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_bo_$o -o bo -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_bo_$o.log 2>&1
done
