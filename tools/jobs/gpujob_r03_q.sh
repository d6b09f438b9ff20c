#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03q; mkdir -p $O; rm -f $O/sweep_720p.txt
for ks in 4,1,4 4,2,4 4,3,4 4,4,4 3,1,4 3,2,4; do
  RMEM_KS=$ks timeout 300 python bench.py --config 720p_k8 --gap 2 --steps 30 --no-cpu-baseline --no-dropin 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('$ks', round(d['value'],1), round(r['mean_us'],1), round(r.get('isolated_mean_us',0),1))" >> $O/sweep_720p.txt
done
