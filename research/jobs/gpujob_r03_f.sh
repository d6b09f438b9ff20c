#!/bin/bash
# round 3, GPU job F: read64 kernel after the issue-priority / tile-iterator / paired-reference changes; which conv is
# not reproducible call to call
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03g; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_ops.py -q -m gpu -k "read_" > $O/pytest_read.log 2>&1; echo "pytest rc $?" >> $O/pytest_read.log
timeout 300 python tools/kbench_read.py --old > $O/kbench_read.json 2> $O/kbench_read.err
timeout 600 python tools/conv_determinism_probe.py > $O/conv_probe.json 2> $O/conv_probe.err
tail -4 $O/pytest_read.log
