cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06f; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "dwconv" 2>&1 | tail -5 | tee $O/pytest_dw.txt
timeout 300 python tools/kbench_dw.py --pk 2>&1 | tail -6 | tee $O/kbench_dw.txt
H=46 W=81 timeout 300 python tools/kbench_dw.py --pk 2>&1 | tail -5 | tee $O/kbench_dw_720p.txt
for f in 2 3; do RMEM_STREAM=$f RMEM_PROJ_KS=$((f==3?4:2)) timeout 300 python tools/kbench_proj.py --trace 2>/dev/null | tail -1 > $O/proj_trace_form$f.json; done
for rep in 1 2; do for pk in 0 1; do
  echo -n "lstt isolated [dw pk $pk] "; RMEM_DW_PK=$pk timeout 300 python tools/lstt_trace.py --replays 200 2>/dev/null | tail -1
done; done 2>&1 | tee $O/lstt_dw.txt
for rep in 1 2; do for pk in 0 1; do
  echo -n "bench [dw pk $pk] "; RMEM_DW_PK=$pk RMEM_BENCH_KERNELS=0 timeout 300 python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), round(r.get('mean_us', 0),1))"
done; done 2>&1 | tee $O/bench_dw.txt
