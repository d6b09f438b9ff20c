#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03q; mkdir -p $O
for rep in 1 2; do
for np_ in 0 1; do
RMEM_NO_PULL=$np_ timeout 600 python bench.py --config clips64 --batched 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('clips64 batched no_pull=$np_', round(d['value'],1), round(d['ms_per_step'],3))"
done
done
