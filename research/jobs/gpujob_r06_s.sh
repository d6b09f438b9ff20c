cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, call s: the last-arriving unit of a query tile merges the key splits (rmem_read_args.tickets): parity, then A/B
O=$PWD/gpurun_out/r06s; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_ops.py -q -m gpu -x -k "read" 2>&1 | tail -5 | tee $O/pytest_read.txt
timeout 1500 python -m pytest tests/test_hip_engine.py tests/test_hip_batched.py -q -m gpu -x 2>&1 | tail -5 | tee $O/pytest_engine.txt
run() { RMEM_MERGE_IN_READ=$1 RMEM_KS=$2 RMEM_BENCH_KERNELS=0 timeout 300 python bench.py --no-cpu-baseline --no-dropin 2>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['roofline'].get('mean_us',0),1))"; }
for rep in 1 2 3; do echo "combine-launch $(run 0 5,1,4)   merge-in-read $(run 1 5,1,4)"; done 2>&1 | tee $O/ab_bench.txt
for ks in 4,1,4 6,1,4 5,1,3 5,1,5 7,1,4; do echo "merge-in-read KS=$ks $(run 1 $ks)"; done 2>&1 | tee $O/ks_sweep.txt
for m in 0 1 0 1; do echo "merge=$m lstt: $(RMEM_MERGE_IN_READ=$m python tools/lstt_trace.py --replays 200 2>/dev/null | tail -1)"; done 2>&1 | tee $O/ab_lstt.txt
for m in 0 1; do echo "merge=$m $(RMEM_MERGE_IN_READ=$m python tools/split_sweep.py 5,1,4 4,1,4 6,1,4 2>&1 | tail -3 | tr '\n' ' ')"; done | tee $O/split_sweep.txt
