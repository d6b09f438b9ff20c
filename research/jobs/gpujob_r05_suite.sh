#!/bin/bash
# the whole GPU suite (no -x: every failure is listed)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r05suite; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu --durations=15 > $O/suite.log 2>&1; echo "suite rc $?" >> $O/suite.log
tail -30 $O/suite.log
