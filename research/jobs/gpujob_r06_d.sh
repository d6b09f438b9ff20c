cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06d; mkdir -p $O
for v in 1 2 3 4; do RMEM_STREAM_VAR=$v timeout 300 python tools/kbench_proj.py --trace 2>/dev/null | tail -1 > $O/proj_trace_var$v.json; done
