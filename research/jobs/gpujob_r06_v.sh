cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, call v: same-box alternating A/B of this tree against the round-5 head (4453889, built under _ab_r05/), then the GPU suite
O=$PWD/gpurun_out/r06v; mkdir -p $O
run() { ( cd $1 && shift && RMEM_BENCH_KERNELS=0 timeout 400 python bench.py --no-cpu-baseline --no-dropin "$@" 2>>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1))" ); }
{
for rep in 1 2 3; do echo "480p K=4 R50-DeAOTL   round-5 head $(run _ab_r05)   this tree $(run .)"; done
for rep in 1 2; do echo "R50-AOTL              round-5 head $(run _ab_r05 --model r50_aotl)   this tree $(run . --model r50_aotl)"; done
for rep in 1 2; do echo "SwinB-AOTL            round-5 head $(run _ab_r05 --model swinb_aotl)   this tree $(run . --model swinb_aotl)"; done
for rep in 1 2; do echo "720p K=8              round-5 head $(run _ab_r05 --config 720p_k8 --gap 2)   this tree $(run . --config 720p_k8 --gap 2)"; done
for rep in 1 2 3; do echo "clips64 batched       round-5 head $(run _ab_r05 --config clips64 --batched)   this tree $(run . --config clips64 --batched)"; done
for rep in 1 2; do echo "8 clips per launch    round-5 head $(run _ab_r05 --batched --clips-per-gpu 8)   this tree $(run . --batched --clips-per-gpu 8)"; done
} 2>&1 | tee $O/ab_vs_round5_head.txt
for rep in 1 2; do
  echo "round-5 head lstt: $(cd _ab_r05 && python tools/lstt_trace.py --replays 200 2>/dev/null | tail -1)"
  echo "this tree   lstt: $(python tools/lstt_trace.py --replays 200 2>/dev/null | tail -1)"
done 2>&1 | tee $O/ab_lstt.txt
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -15 | tee $O/pytest_gpu.log
