#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04encb; mkdir -p $O
for rep in 1 2; do
for b in 2 3 4; do
  echo -n "RMEM_ENC_BATCH=$b: "; RMEM_ENC_BATCH=$b timeout 300 python bench.py --steps 240 --no-cpu-baseline --no-dropin 2>> $O/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['value'],1), round(d['roofline']['mean_us'],1))"
done; done | tee $O/r04_enc_batch_sweep.txt
