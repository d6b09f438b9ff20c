#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03v; mkdir -p $O
for i in 1 2; do timeout 600 python -m pytest tests/test_hip_ops.py -q -m gpu -k "read_" > $O/pytest_read_$i.log 2>&1; tail -1 $O/pytest_read_$i.log; done
timeout 300 python tools/kbench_read.py > $O/kbench_read.json 2> $O/kbench_read.err
timeout 300 python tools/kbench.py > $O/kbench.json 2> $O/kbench.err
