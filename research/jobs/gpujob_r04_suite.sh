#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04suite; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --timeout 400 -x 2>&1 | tail -30 > $O/gpu_tests.log
tail -12 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 400 python bench.py 2> $O/bench.err | tee $O/r04_bench_final.json | cut -c1-400
