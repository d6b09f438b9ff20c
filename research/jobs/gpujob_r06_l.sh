cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, call l: full GPU suite on the cleaned tree, then AOT bench with the first streaming form (generic shapes inside it) vs the second (generic shapes -> tile kernels)
O=gpurun_out/r06l; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu -x -s 2>&1 | tee $O/pytest_gpu_full.log | tail -6
for rep in 1 2; do for f in 1 2; do
  echo -n "bench r50_aotl [stream form $f] "; RMEM_STREAM=$f RMEM_BENCH_KERNELS=0 timeout 400 python bench.py --model r50_aotl --no-cpu-baseline --no-dropin 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), round(r.get('mean_us', 0),1))"
done; done 2>&1 | tee $O/bench_aot_forms.txt
for rep in 1 2; do for f in 1 2; do
  echo -n "bench default [stream form $f] "; RMEM_STREAM=$f RMEM_BENCH_KERNELS=0 timeout 400 python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), round(r.get('mean_us', 0),1))"
done; done 2>&1 | tee $O/bench_forms.txt
