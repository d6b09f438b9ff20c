#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04c64; mkdir -p $O
for rep in 1 2; do
  echo -n "tiles: "; RMEM_LINEAR=tiles timeout 900 python bench.py --config clips64 --batched 2>> $O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],1))"
  echo -n "default: "; timeout 900 python bench.py --config clips64 --batched 2>> $O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],1))"
done | tee $O/r04_clips64_batched_order.txt
