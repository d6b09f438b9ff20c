#!/bin/bash
# round 5, first GPU call: the fused LayerNorm + grouped projection kernel (linear_rowres.h) -- parity, then isolated LSTT A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "ln_linear or single_stage or linear_stream or layernorm" > $O/ops.log 2>&1; echo "ops rc $?" >> $O/ops.log
tail -5 $O/ops.log
timeout 900 python -m pytest tests/test_hip_engine.py -x -q -m gpu -k "lstt_forward or small_clip" > $O/eng.log 2>&1; echo "eng rc $?" >> $O/eng.log
tail -5 $O/eng.log
for i in 1 2; do
  RMEM_ROWRES=0 timeout 300 python tools/lstt_trace.py >> $O/lstt_old.json 2>> $O/err.log
  timeout 300 python tools/lstt_trace.py >> $O/lstt_new.json 2>> $O/err.log
done
cat $O/lstt_old.json $O/lstt_new.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o lstt -- python tools/lstt_trace.py > $O/lstt_prof.json 2>> $O/err.log
python tools/prof_summary.py $O/prof/lstt_kernel_trace.csv 30 > $O/r05a_lstt_isolated_kernel_stats.md 2>> $O/err.log
head -30 $O/r05a_lstt_isolated_kernel_stats.md | cut -c1-200
find $O/prof -name "*.csv" -size +1M -delete
tail -5 $O/err.log
