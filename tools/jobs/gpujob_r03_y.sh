#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03y; mkdir -p $O; rm -f $O/sweep3.txt
for ks in 7,2,6 8,3,6 8,4,6 9,3,6 9,4,6 10,4,6 8,5,6; do
  RMEM_KS=$ks timeout 120 python tools/kbench.py --only reads 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('$ks', d['read2_long+window'], d['read_combine2'])" >> $O/sweep3.txt
done
for ks in 7,2,6 8,3,6 8,4,6 9,4,6; do
  RMEM_KS=$ks timeout 300 python bench.py --steps 80 --no-cpu-baseline --no-dropin 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('bench $ks', round(d['value'],1), round(d['roofline']['mean_us'],1))" >> $O/sweep3.txt
done
