cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, call j: key splits in the frame with single-split reads gating their own output
O=gpurun_out/r06j; mkdir -p $O
for rep in 1 2; do for ks in 7,2,4 7,1,4 6,1,4 5,1,4 5,1,3 4,1,4; do echo -n "RMEM_KS=$ks "; RMEM_KS=$ks RMEM_BENCH_KERNELS=0 timeout 300 python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), round(r.get('mean_us', 0),1))"; done; done 2>&1 | tee $O/ks_sweep_fused.txt
echo -n "RMEM_KS=5,1,4 RMEM_FUSE_GATE=0 "; RMEM_FUSE_GATE=0 RMEM_KS=5,1,4 RMEM_BENCH_KERNELS=0 timeout 300 python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), round(r.get('mean_us', 0),1))" | tee -a $O/ks_sweep_fused.txt
python tools/split_sweep.py 7,2,4 7,1,4 6,1,4 5,1,4 2>&1 | tee $O/split_sweep_isolated.txt
for ks in 7,2,4 5,1,4; do echo -n "lstt isolated [$ks] "; RMEM_KS=$ks timeout 300 python tools/lstt_trace.py --replays 200 2>/dev/null | tail -1; done | tee $O/lstt_ks.txt
