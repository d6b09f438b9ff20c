#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03q; mkdir -p $O; rm -f $O/b4_steps.txt
for st in 40 100 100 60 80; do
timeout 900 python bench.py --batched --clips-per-gpu 4 --steps $st --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('steps=$st', round(d['value'],1), round(d['ms_per_step'],2), d['config'].get('memory_path_launches_per_step'))" >> $O/b4_steps.txt
done
