#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04queue; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_batched.py -x -q -m gpu -s 2>&1 | tail -60 > $O/batched_tests.log
tail -25 $O/batched_tests.log
timeout 2500 python -m pytest tests -x -q -m gpu --deselect tests/test_hip_batched.py 2>&1 | tail -15 > $O/gpu_tests.log
tail -8 $O/gpu_tests.log
