cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, call q: band-order relative bias (no gathers in the windowed score phase): op tests, read timings, splits in the frame
O=gpurun_out/r06q; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_ops.py -q -m gpu -x 2>&1 | tail -5 | tee $O/pytest_ops.txt
python tools/split_sweep.py 5,1,4 6,1,4 7,1,4 7,2,4 8,1,4 2>&1 | tee $O/split_sweep_isolated.txt
for rep in 1 2; do for ks in 5,1,4 6,1,4 7,1,4 7,2,4 8,1,4; do echo -n "RMEM_KS=$ks "; RMEM_KS=$ks RMEM_BENCH_KERNELS=0 timeout 300 python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), round(r.get('mean_us', 0),1))"; done; done 2>&1 | tee $O/ks_sweep_band.txt
for ks in 5,1,4 6,1,4 7,1,4; do echo -n "lstt isolated [$ks] "; RMEM_KS=$ks timeout 300 python tools/lstt_trace.py --replays 200 2>/dev/null | tail -1; done | tee $O/lstt_ks.txt
