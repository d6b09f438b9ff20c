#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04y; mkdir -p $O
timeout 1500 python -m pytest tests/test_driver.py -q -m gpu -x -k "rccl or world1 or clips64" > $O/driver_tests.log 2>&1; tail -4 $O/driver_tests.log
timeout 600 python tools/kbench.py > $O/r04y_kbench.json 2> $O/kb.err; cat $O/r04y_kbench.json | tr '\n' ' '; echo; tail -2 $O/kb.err
