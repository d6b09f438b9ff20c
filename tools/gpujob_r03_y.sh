#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03y; mkdir -p $O; rm -f $O/sweep_bench2.txt
for rep in 1 2; do
for ks in 7,2,6 7,2,9 7,2,8 7,2,12; do
  RMEM_KS=$ks timeout 300 python bench.py --steps 80 --no-cpu-baseline --no-dropin 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$ks', round(d['value'],1), round(d['roofline']['mean_us'],1))" >> $O/sweep_bench2.txt
done
done
