#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04c64; mkdir -p $O
for rep in 1 2 3; do
  echo -n "clips64 batched: "; timeout 900 python bench.py --config clips64 --batched 2>> $O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), d['config']['per_rank_seconds'])"
done | tee $O/r04_clips64_batched_repeats.txt
echo -n "clips64 batched, tiles+KS as r04m (RMEM_LINEAR=tiles): "; RMEM_LINEAR=tiles timeout 900 python bench.py --config clips64 --batched 2>> $O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],1))" | tee -a $O/r04_clips64_batched_repeats.txt
