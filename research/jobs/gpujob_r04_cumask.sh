#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04cumask; mkdir -p $O
timeout 120 python research/debug/cumask_probe.py 2>&1 | grep -v amdgpu | tee $O/cumask_probe.txt
for rep in 1 2; do
for n in 0 192 160 128 96; do
  echo -n "RMEM_ENC_CUS=$n: "; RMEM_ENC_CUS=$n timeout 200 python bench.py --steps 240 --no-cpu-baseline --no-dropin 2>> $O/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['value'],1), round(d['roofline']['mean_us'],1))"
done; done | tee $O/r04_enc_cu_mask_sweep.txt
