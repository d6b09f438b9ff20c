cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, call z: AOT block with shared launches (RMEM_AOT_GROUP): bit-identity, then A/B on R50-AOTL and SwinB-AOTL
O=$PWD/gpurun_out/r06z; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_aot.py -q -m gpu -x 2>&1 | tail -12 | tee $O/pytest_aot.txt
run() { RMEM_AOT_GROUP=$1 RMEM_BENCH_KERNELS=0 timeout 400 python bench.py --model $2 --no-cpu-baseline --no-dropin 2>>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1))"; }
for rep in 1 2 3; do echo "R50-AOTL   one launch each $(run 0 r50_aotl)   shared launches $(run 1 r50_aotl)"; done 2>&1 | tee $O/ab_aot.txt
for rep in 1 2; do echo "SwinB-AOTL one launch each $(run 0 swinb_aotl)   shared launches $(run 1 swinb_aotl)"; done 2>&1 | tee -a $O/ab_aot.txt
