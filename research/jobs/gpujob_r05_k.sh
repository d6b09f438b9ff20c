#!/bin/bash
# depth-wise conv: RY rows per thread -- parity, isolated timing, in-frame A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r05k; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -q -m gpu -x -k "dwconv or groupnorm2_fold" 2>&1 | tail -2
timeout 300 python tools/kbench_dw.py --rows 2>/dev/null | tee $O/kbench_dw_480p.txt
H=46 W=81 timeout 300 python tools/kbench_dw.py --rows 2>/dev/null | tee $O/kbench_dw_720p.txt
for i in 1 2; do
  for ry in 0 2 3 4; do
    RMEM_DW_ROWS=$ry timeout 600 python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
k = {e['kernel'][:6]: round(e['us_per_frame']) for e in d['roofline']['kernels']}
print('RMEM_DW_ROWS=$ry', round(d['value'], 1), 'fps', k)" | tee -a $O/bench_ab_dw_rows.txt
  done
done
