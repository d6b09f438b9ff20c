#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03y; mkdir -p $O; rm -f $O/sweep_sched.txt
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 80 --no-cpu-baseline --no-dropin 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label', round(d['value'],1), round(d['roofline']['mean_us'],1))" >> $O/sweep_sched.txt
}
for rep in 1 2; do
run default X=1
run enc1 RMEM_ENC_BATCH=1
run pf_dec RMEM_PREFETCH_AT=decoder
run hoist0 RMEM_HOIST=0
run enc1_hoist0 RMEM_ENC_BATCH=1 RMEM_HOIST=0
run unpaired RMEM_BRANCH_ORDER=serial_unpaired
done
