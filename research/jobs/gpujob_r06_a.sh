cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, call a: baseline of this box, kernel arguments in device memory (HIP_FORCE_DEV_KERNARG), window read in one split
O=gpurun_out/r06a; mkdir -p $O
for rep in 1 2; do
for e in "X=0" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0"; do
  echo -n "lstt isolated [$e] " ; env $e timeout 300 python tools/lstt_trace.py --replays 200 2>/dev/null | tail -1
done; done 2>&1 | tee $O/lstt_kernarg.txt
for rep in 1 2; do
for e in "X=0" "HIP_FORCE_DEV_KERNARG=1"; do
  echo -n "bench [$e] "; env $e RMEM_BENCH_KERNELS=0 timeout 300 python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), round(r.get('mean_us', 0),1))"
done; done 2>&1 | tee $O/bench_kernarg.txt
for ks in 7,2,4 7,1,4 8,1,4 6,1,4 5,1,4 5,1,3 7,2,4; do echo -n "RMEM_KS=$ks "; RMEM_KS=$ks RMEM_BENCH_KERNELS=0 timeout 300 python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), round(r.get('mean_us', 0),1))"; done 2>&1 | tee $O/ks_sweep.txt
python tools/split_sweep.py 7,2,4 7,1,4 8,1,4 6,1,4 5,1,4 2>&1 | tee $O/split_sweep_isolated.txt
python tools/kbench_gemm.py 2>/dev/null | tee $O/kbench_gemm.json
python tools/kbench_gemm.py --trace 2>/dev/null | tee $O/stream_trace.json
