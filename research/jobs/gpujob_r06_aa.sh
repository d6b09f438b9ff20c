cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, call aa: key splits of the AOT block's reads under the shared launches
O=$PWD/gpurun_out/r06aa; mkdir -p $O
run() { RMEM_AOT_KS=$1 RMEM_BENCH_KERNELS=0 timeout 400 python bench.py --model r50_aotl --no-cpu-baseline --no-dropin 2>>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1))"; }
for rep in 1 2; do for ks in 6,4 5,3 4,2 8,4 6,2 4,4 3,2; do echo "RMEM_AOT_KS=$ks $(run $ks)"; done; done 2>&1 | tee $O/aot_ks_sweep_paired.txt
