#!/bin/bash
# round 3, GPU job B: the 64-query read kernel -- op tests, old-vs-new timing with phase stamps, rerun determinism
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03e; mkdir -p $O
export TMPDIR=/tmp
echo skip > $O/pytest_read.log
timeout 300 python tools/kbench_read.py --old > $O/kbench_read.json 2> $O/kbench_read.err
timeout 300 python tools/rerun_determinism_probe.py > $O/rerun_probe.log 2>&1
tail -15 $O/pytest_read.log
