#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03q; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_batched.py -q -m gpu -k "window or ragged" 2>&1 | tail -8
