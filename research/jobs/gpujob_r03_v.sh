#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03v; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py -q -m gpu -k "read_" > $O/pytest_read_1.log 2>&1; tail -1 $O/pytest_read_1.log
timeout 300 python tools/kbench.py --only reads > $O/kbench.json 2> $O/kbench.err
timeout 300 python tools/kbench_read.py > $O/kbench_read.json 2> $O/kbench_read.err
