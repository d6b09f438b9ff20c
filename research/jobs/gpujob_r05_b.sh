#!/bin/bash
# round 5: DPP butterfly in the shared LayerNorm row function; fused kernel stamps + isolated LSTT A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r05b; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "ln_linear or layernorm or linear_stream" > $O/ops.log 2>&1; echo "ops rc $?" >> $O/ops.log
tail -3 $O/ops.log
timeout 300 python tools/kbench_rowres.py --trace > $O/r05b_kbench_rowres.json 2>> $O/err.log
cat $O/r05b_kbench_rowres.json
for i in 1 2; do
  RMEM_ROWRES=0 timeout 300 python tools/lstt_trace.py >> $O/lstt_old.json 2>> $O/err.log
  timeout 300 python tools/lstt_trace.py >> $O/lstt_new.json 2>> $O/err.log
done
cat $O/lstt_old.json $O/lstt_new.json
