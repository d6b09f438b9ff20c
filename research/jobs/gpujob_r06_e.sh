cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, call e: 128 x 128 form of the streaming kernel (RMEM_STREAM=3): bit-identity, per-launch timings, stamps, LSTT, bench
O=gpurun_out/r06e; mkdir -p $O
RMEM_STREAM=3 timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "linear or ln_linear or layernorm or lstt" 2>&1 | tail -8 | tee $O/pytest_linear_form3.txt
for cfg in "2 2" "3 2" "3 4" "2 2" "3 2" "3 4"; do set -- $cfg
  echo -n "form $1 KS $2: "; RMEM_STREAM=$1 RMEM_PROJ_KS=$2 timeout 300 python tools/kbench_proj.py 2>$O/kproj_err.txt | tail -1
done | tee $O/kbench_proj.txt
RMEM_STREAM=3 RMEM_PROJ_KS=4 timeout 300 python tools/kbench_proj.py --trace 2>/dev/null | tail -1 > $O/proj_trace_form3_ks4.json
RMEM_STREAM=3 RMEM_PROJ_KS=2 timeout 300 python tools/kbench_proj.py --trace 2>/dev/null | tail -1 > $O/proj_trace_form3_ks2.json
for rep in 1 2 3; do for cfg in "2 2" "3 2" "3 4"; do set -- $cfg
  echo -n "lstt isolated [form $1 KS $2] "; RMEM_STREAM=$1 RMEM_PROJ_KS=$2 timeout 300 python tools/lstt_trace.py --replays 200 2>/dev/null | tail -1
done; done 2>&1 | tee $O/lstt_forms.txt
for rep in 1 2; do for cfg in "2 2" "3 2" "3 4"; do set -- $cfg
  echo -n "bench [form $1 KS $2] "; RMEM_STREAM=$1 RMEM_PROJ_KS=$2 RMEM_BENCH_KERNELS=0 timeout 300 python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), round(r.get('mean_us', 0),1))"
done; done 2>&1 | tee $O/bench_forms.txt
