cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, call k: ops + engine tests after the cleanup (rmem_configure, rowres removed, fused gate), then the split sweep in the frame
O=gpurun_out/r06k; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_ops.py tests/test_hip_batched.py -q -m gpu -x 2>&1 | tail -4 | tee $O/pytest_ops.txt
bash research/jobs/gpujob_r06_j.sh
