cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, call r: same-box A/B of the band-order relative bias against the tree before it (_ab_head = git HEAD, built there)
O=$PWD/gpurun_out/r06r; mkdir -p $O
run() { ( cd $1 && RMEM_BENCH_KERNELS=0 timeout 300 python bench.py --no-cpu-baseline --no-dropin 2>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))" ); }
for rep in 1 2 3 4; do echo "head $(run _ab_head)  band $(run .)"; done 2>&1 | tee $O/ab_bench.txt
for rep in 1 2; do
  echo "head window: $(cd _ab_head && python tools/kbench_read.py --only window --no-trace 2>&1 | tail -3 | tr '\n' ' ')"
  echo "band window: $(python tools/kbench_read.py --only window --no-trace 2>&1 | tail -3 | tr '\n' ' ')"
done 2>&1 | tee $O/ab_window_read.txt
for rep in 1 2; do
  echo "head lstt: $(cd _ab_head && python tools/lstt_trace.py --replays 200 2>/dev/null | tail -1)"
  echo "band lstt: $(python tools/lstt_trace.py --replays 200 2>/dev/null | tail -1)"
done 2>&1 | tee $O/ab_lstt.txt
