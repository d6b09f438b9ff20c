cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, call p: frame-schedule knobs re-swept under the round's kernels and splits (same box, alternating)
O=gpurun_out/r06p; mkdir -p $O
b() { echo -n "$* : "; env "$@" RMEM_BENCH_KERNELS=0 timeout 300 python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), round(r.get('mean_us', 0),1))"; }
for rep in 1 2; do
b X=0
b RMEM_ENC_BATCH=1
b RMEM_ENC_BATCH=3
b RMEM_ENC_BATCH=4
b RMEM_PREFETCH_AT=decoder
b RMEM_HOIST=0
b RMEM_EARLY_LONG_READ=1
b RMEM_PROJ_KS=1
b RMEM_PROJ_KS=3
b RMEM_DW_ROWS=3
done 2>&1 | tee $O/schedule_knobs.txt
