cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, call c: second form with scalar-cache warm-up and early exit; per-launch timings; padded leading dimensions; MFMA/VALU interleave ubench
O=gpurun_out/r06c; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "linear or ln_linear or layernorm or lstt" 2>&1 | tail -5 | tee $O/pytest_linear.txt
for f in 1 2 1 2; do
  echo -n "form $f: "; RMEM_STREAM=$f timeout 300 python tools/kbench_proj.py 2>$O/kproj_err_$f.txt | tail -1
done | tee $O/kbench_proj.txt
for f in 1 2; do RMEM_STREAM=$f timeout 300 python tools/kbench_proj.py --trace 2>/dev/null | tail -1 > $O/proj_trace_form$f.json; done
for rep in 1 2 3; do for f in 1 2; do
  echo -n "lstt isolated [form $f] "; RMEM_STREAM=$f timeout 300 python tools/lstt_trace.py --replays 200 2>/dev/null | tail -1
done; done 2>&1 | tee $O/lstt_forms.txt
timeout 300 research/ubench/mfma_valu_interleave 2>&1 | tee $O/mfma_valu_interleave.txt
