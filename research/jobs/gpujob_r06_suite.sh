cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06suite; mkdir -p $O
timeout 3400 python -m pytest tests -q -m gpu -x -s 2>&1 | tee $O/pytest_gpu_full.log | tail -40
