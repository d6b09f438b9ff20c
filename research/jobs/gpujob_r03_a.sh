#!/bin/bash
# round 3, GPU job A: full -m gpu suite, baseline kernel numbers (+ phase traces incl. the windowed read),
# process-level determinism probe of the MIOpen stages.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03a; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
timeout 300 python tools/kbench.py > $O/kbench.json 2> $O/kbench.err
timeout 300 python tools/kbench_read.py --splits 6,7,8 > $O/kbench_read.json 2> $O/kbench_read.err
timeout 300 python tools/kbench_read.py --splits 1,2,3 --only window > $O/kbench_read_win.json 2>> $O/kbench_read.err
timeout 900 python tools/parity_mode_probe.py > $O/parity_mode_probe.json 2> $O/parity_mode_probe.err
tail -5 $O/pytest_gpu.log
