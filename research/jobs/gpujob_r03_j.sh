#!/bin/bash
# round 3, GPU job J: sampled reference pass; same-box A/B of the headline bench (round-2 kernel / read64 / read64 without
# MIOpen's implicit-GEMM solvers); the two rerun tests with that switch
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03j; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_ops.py -q -m gpu -k "read_" > $O/pytest_read.log 2>&1; echo "pytest rc $?" >> $O/pytest_read.log
timeout 300 python tools/kbench_read.py > $O/kbench_read.json 2> $O/kbench_read.err
timeout 300 python tools/kbench.py --only reads > $O/kbench_reads.json 2> $O/kbench.err
for v in new v128 new_noigemm; do
  if [ $v = v128 ]; then export RMEM_READ_IMPL=v128; else unset RMEM_READ_IMPL; fi
  if [ $v = new_noigemm ]; then export MIOPEN_DEBUG_CONV_IMPLICIT_GEMM=0; else unset MIOPEN_DEBUG_CONV_IMPLICIT_GEMM; fi
  timeout 400 python bench.py --no-cpu-baseline --no-dropin --steps 200 > $O/bench_$v.json 2> $O/bench_$v.err
done
MIOPEN_DEBUG_CONV_IMPLICIT_GEMM=0 timeout 600 python -m pytest tests/test_hip_batched.py -q -m gpu -k "load_network" > $O/pytest_rerun.log 2>&1
tail -3 $O/pytest_read.log $O/pytest_rerun.log; for v in new v128 new_noigemm; do python -c "
import json,sys
d=json.load(open('$O/bench_$v.json')); print('$v', round(d['value'],1), round(d['roofline']['mean_us'],1), round(d['roofline']['isolated_mean_us'],1))"; done
