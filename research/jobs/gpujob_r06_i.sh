cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06i; mkdir -p $O
for f in 2 1; do echo "== RMEM_STREAM=$f"; RMEM_STREAM=$f timeout 900 python -m pytest tests/test_driver.py -q -m gpu -k "test_driver_hip_vs_reference_golden" 2>&1 | tail -6; done | tee $O/bisect_stream.txt
echo "== RMEM_FUSE_GATE=0"; RMEM_FUSE_GATE=0 timeout 900 python -m pytest tests/test_driver.py -q -m gpu -k "test_driver_hip_vs_reference_golden" 2>&1 | tail -4 | tee -a $O/bisect_stream.txt
timeout 900 python -m pytest tests/test_hip_ops.py -q -m gpu -k "single_split or read_window or read_bank" 2>&1 | tail -4 | tee $O/pytest_fuse.txt
