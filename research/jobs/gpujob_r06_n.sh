cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, call n: the driver tests again (8-rank pinned), then AOT key splits in the frame
O=gpurun_out/r06n; mkdir -p $O
timeout 2400 python -m pytest tests/test_driver.py -q -m gpu -x 2>&1 | tail -5 | tee $O/pytest_driver.txt
for rep in 1 2; do for ks in 12,12 8,8 8,4 6,4 12,4 16,6; do echo -n "RMEM_AOT_KS=$ks "; RMEM_AOT_KS=$ks RMEM_BENCH_KERNELS=0 timeout 400 python bench.py --model r50_aotl --no-cpu-baseline --no-dropin 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), round(r.get('mean_us', 0),1))"; done; done 2>&1 | tee $O/aot_ks_sweep.txt
