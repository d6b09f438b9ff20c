#!/bin/bash
# final check of the round: whole GPU suite, smoke(), default bench on the committed kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04zz; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/r04zz_bench_x3.json 2> $O/bench_x3.err; head -c 1800 $O/r04zz_bench_x3.json; echo; tail -2 $O/bench_x3.err
