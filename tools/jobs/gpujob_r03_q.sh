#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03q; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_engine.py tests/test_hip_batched.py -q -m gpu -x -k "unit_queue or paired or batched" > $O/pytest_pull.log 2>&1; tail -2 $O/pytest_pull.log
for b in 8 2; do
timeout 600 python bench.py --batched --clips-per-gpu $b --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('batched $b', round(d['value'],1), round(d['ms_per_step'],2), round(r['mean_us'],1), round(r['frac'],4))"
done
for np_ in 0 1; do
RMEM_NO_PULL=$np_ timeout 300 python bench.py --config 720p_k8 --gap 2 --steps 30 --no-cpu-baseline --no-dropin 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('720p no_pull=$np_', round(d['value'],1), round(r['mean_us'],1), round(r.get('isolated_mean_us',0),1))"
done
