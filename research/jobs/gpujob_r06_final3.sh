cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, final tree: the GPU suite
O=$PWD/gpurun_out/r06final; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -12 | tee $O/pytest_gpu.log
