#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03q; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_batched.py -q -m gpu -x > $O/pytest_batched.log 2>&1; tail -1 $O/pytest_batched.log
for b in 8 4 2; do
timeout 600 python bench.py --batched --clips-per-gpu $b --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('batched $b', round(d['value'],1), round(d['ms_per_step'],2), round(r['mean_us'],1), round(r['frac'],4))"
done
