cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r06x; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_batched.py -q -m gpu -s -k "mid_clip" 2>&1 | tail -12 | tee $O/pytest.txt
