cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r06ag; mkdir -p $O
timeout 2400 python -m pytest tests/test_driver.py -q -m gpu 2>&1 | tail -6 | tee $O/pytest_driver.txt
