cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, call w: split-per-XCD layout of the paired read (rmem_configure read_xcd): bit-identity, isolated and in-frame A/B
O=$PWD/gpurun_out/r06w; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_engine.py -q -m gpu -k "split_per_xcd or paired or unit_queue" 2>&1 | tail -6 | tee $O/pytest.txt
for x in 0 1 0 1; do echo "read_xcd=$x $(RMEM_READ_XCD=$x python tools/split_sweep.py 5,1,4 2>&1 | tail -1)"; done | tee $O/split_sweep.txt
run() { RMEM_READ_XCD=$1 RMEM_KS=$2 RMEM_BENCH_KERNELS=0 timeout 300 python bench.py --no-cpu-baseline --no-dropin 2>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['roofline'].get('mean_us',0),1))"; }
for rep in 1 2 3; do echo "chunked $(run 0 5,1,4)   split-per-XCD $(run 1 5,1,4)"; done 2>&1 | tee $O/ab_bench.txt
for ks in 4,1,4 6,1,4 7,1,4; do echo "split-per-XCD KS=$ks $(run 1 $ks)   chunked $(run 0 $ks)"; done 2>&1 | tee $O/ks_sweep.txt
for x in 0 1 0 1; do echo "read_xcd=$x lstt: $(RMEM_READ_XCD=$x python tools/lstt_trace.py --replays 200 2>/dev/null | tail -1)"; done 2>&1 | tee $O/ab_lstt.txt
