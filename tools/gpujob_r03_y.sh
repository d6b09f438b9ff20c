#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03y; mkdir -p $O
for ks in 6,2,6 7,2,6 8,2,6 9,2,6 7,1,6 8,1,6 9,1,6 7,3,6 6,3,6 7,2,4 7,2,9 8,1,9 5,2,6; do
  RMEM_KS=$ks timeout 120 python tools/kbench.py --only reads 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('$ks', {k:v for k,v in d.items() if 'read' in k or 'lstt' in k})" >> $O/sweep_kbench.txt
done
for ks in 7,2,6 8,2,6 9,2,6 8,1,9 9,1,9 6,2,6; do
  RMEM_KS=$ks timeout 300 python bench.py --steps 60 --no-cpu-baseline --no-dropin 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$ks', round(d['value'],1), round(d['roofline']['mean_us'],1))" >> $O/sweep_bench.txt
done
