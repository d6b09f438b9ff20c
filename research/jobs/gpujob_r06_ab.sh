cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, call ab: AOT-block clips in flight on separate streams (bench.py --clips-per-gpu)
O=$PWD/gpurun_out/r06ab; mkdir -p $O
for n in 1 2 3 4; do echo "R50-AOTL clips-per-gpu=$n $(RMEM_BENCH_KERNELS=0 timeout 600 python bench.py --model r50_aotl --clips-per-gpu $n --no-cpu-baseline --no-dropin 2>>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['config'].get('parallelism'))")"; done 2>&1 | tee $O/aot_clips_in_flight.txt
for n in 2 3; do echo "R50-DeAOTL clips-per-gpu=$n $(RMEM_BENCH_KERNELS=0 timeout 600 python bench.py --clips-per-gpu $n --no-cpu-baseline --no-dropin 2>>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1))")"; done 2>&1 | tee -a $O/aot_clips_in_flight.txt
tail -5 $O/err.txt
