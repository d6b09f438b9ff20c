#!/bin/bash
# last check of the round: whole GPU suite, smoke(), default bench on the committed kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03z; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/r03z_bench_x3.json 2> $O/bench_x3.err; tail -c 400 $O/r03z_bench_x3.json
