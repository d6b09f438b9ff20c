#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03y; mkdir -p $O; rm -f $O/sweep4.txt
for ks in 7,2,6 8,2,6 8,3,6 8,4,6 9,3,6 9,4,6 10,3,6 10,4,6 12,4,6; do
  RMEM_KS=$ks timeout 120 python tools/kbench.py --only reads 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('480p $ks', d['read2_long+window'], d['read_combine2'])" >> $O/sweep4.txt
done
for ks in 7,2,6 8,3,6 9,3,6 9,4,6; do
  RMEM_KS=$ks timeout 300 python bench.py --steps 80 --no-cpu-baseline --no-dropin 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('480p bench $ks', round(d['value'],1), round(d['roofline']['mean_us'],1))" >> $O/sweep4.txt
done
for ks in 4,2,4 5,2,4 6,2,4 5,3,4 6,3,4 8,3,4; do
  RMEM_KS=$ks timeout 300 python bench.py --config 720p_k8 --gap 2 --steps 30 --no-cpu-baseline --no-dropin 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('720p bench $ks', round(d['value'],1), round(r['mean_us'],1))" >> $O/sweep4.txt
done
