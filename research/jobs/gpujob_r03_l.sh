#!/bin/bash
# round 3, GPU job L: register-staged reference pass, device-side eviction: full suite, kbench, bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03l; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_ops.py -q -m gpu -k "read_ or fg_weights or bank_policy" > $O/pytest_ops.log 2>&1; echo "pytest rc $?" >> $O/pytest_ops.log
timeout 300 python tools/kbench.py > $O/kbench.json 2> $O/kbench.err
timeout 300 python tools/kbench_read.py > $O/kbench_read.json 2> $O/kbench_read.err
timeout 1500 python -m pytest tests -q -m gpu --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
RMEM_HOST_POLICY=1 timeout 600 python bench.py --no-cpu-baseline --no-dropin > $O/bench_host_policy.json 2>> $O/bench.err
tail -4 $O/pytest_ops.log; tail -6 $O/pytest_gpu.log; head -c 600 $O/bench.json
