#!/bin/bash
# round 3, GPU job K: full -m gpu suite after the kernel clean-up (old kernel removed, implicit GEMM off by default),
# per-kernel bench, headline bench with rocprofv3 kernel stats
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03k; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --durations=10 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
timeout 300 python tools/kbench.py > $O/kbench.json 2> $O/kbench.err
timeout 300 python tools/kbench_read.py > $O/kbench_read.json 2> $O/kbench_read.err
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -6 $O/pytest_gpu.log; head -c 900 $O/bench.json
