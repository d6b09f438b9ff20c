cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, call g: software-pipelined mha_flash: bit-identity against the serial kernel, timings, AOT bench A/B
O=gpurun_out/r06g; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_aot.py -x -q -m gpu -k "mha_flash" 2>&1 | tail -8 | tee $O/pytest_mha.txt
for f in serial pipe serial pipe; do echo -n "$f: "; RMEM_MHA=$f timeout 300 python tools/kbench_mha.py --splits 8,12,16 2>/dev/null | tail -1; done | tee $O/kbench_mha.txt
for rep in 1 2; do for f in serial pipe; do
  echo -n "bench r50_aotl [$f] "; RMEM_MHA=$f RMEM_BENCH_KERNELS=0 timeout 400 python bench.py --model r50_aotl --no-cpu-baseline --no-dropin 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), round(r.get('mean_us', 0),1))"
done; done 2>&1 | tee $O/bench_aot.txt
