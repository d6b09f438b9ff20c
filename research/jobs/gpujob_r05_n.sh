cd "${GRAFT_REPO_ROOT:-/root/repo}"
# in-frame sweep of the key splits (long, window, self) of the attention reads: frames/s of the default bench
for ks in ${KS_LIST:-7,2,6 5,2,4 5,2,3 5,2,2 6,2,3 5,1,3 3,1,2 6,1,4 7,2,6 5,2,4 5,2,3}; do echo -n "RMEM_KS=$ks "; RMEM_KS=$ks RMEM_BENCH_KERNELS=0 timeout 300 python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), round(r.get('mean_us', 0),1))"; done
