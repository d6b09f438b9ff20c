cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, call b: second form of the streaming projection kernel -- bit-identity tests, per-launch timings, stamps, LSTT, bench A/B
O=gpurun_out/r06b; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "linear or ln_linear or layernorm or lstt" 2>&1 | tail -15 | tee $O/pytest_linear.txt
for f in 1 2; do
  echo "== RMEM_STREAM=$f"; RMEM_STREAM=$f timeout 300 python tools/kbench_gemm.py 2>$O/kbench_err_$f.txt | tee $O/kbench_gemm_form$f.json
  RMEM_STREAM=$f timeout 300 python tools/kbench_gemm.py --trace 2>/dev/null | tee $O/stream_trace_form$f.json
done
for rep in 1 2 3; do for f in 1 2; do
  echo -n "lstt isolated [form $f] "; RMEM_STREAM=$f timeout 300 python tools/lstt_trace.py --replays 200 2>/dev/null | tail -1
done; done 2>&1 | tee $O/lstt_forms.txt
for rep in 1 2 3; do for f in 1 2; do
  echo -n "bench [form $f] "; RMEM_STREAM=$f RMEM_BENCH_KERNELS=0 timeout 300 python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), round(r.get('mean_us', 0),1))"
done; done 2>&1 | tee $O/bench_forms.txt
