#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04multi; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_engine.py -x -q -m gpu -s -k "multi_object or grows_past" --timeout 400 2>&1 | tail -30 > $O/tests.log
tail -12 $O/tests.log | cut -c1-400
timeout 600 python tools/multiobj_bench.py 2> $O/mb.err | tee $O/r04_multiobj_bench.json
tail -5 $O/mb.err
