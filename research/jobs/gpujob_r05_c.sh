#!/bin/bash
# round 5: row-resident projection kernel, three ways of running a layer front (old / LN launch + planes / LN fused)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r05c; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "ln_linear or layernorm or linear_stream" > $O/ops.log 2>&1; echo "ops rc $?" >> $O/ops.log
tail -3 $O/ops.log
timeout 300 python tools/kbench_rowres.py --trace > $O/r05c_kbench_rowres.json 2>> $O/err.log
cat $O/r05c_kbench_rowres.json
for i in 1 2; do
  for m in 0 planes fused; do
    echo -n "$m " >> $O/lstt_modes.txt
    RMEM_ROWRES=$m timeout 300 python tools/lstt_trace.py >> $O/lstt_modes.txt 2>> $O/err.log
  done
done
cat $O/lstt_modes.txt
timeout 900 python -m pytest tests/test_hip_engine.py -x -q -m gpu -k "lstt_forward or small_clip" > $O/eng.log 2>&1; echo "eng rc $?" >> $O/eng.log
tail -3 $O/eng.log
