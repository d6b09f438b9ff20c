#!/bin/bash
# per-kernel isolated durations of the LSTT graph: round-4 projection path (RMEM_ROWRES=0) against the fused row-resident launch
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r05d; mkdir -p $O
for m in 0 fused planes; do
  RMEM_ROWRES=$m timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -o lstt -- python tools/lstt_trace.py > $O/lstt_$m.json 2>> $O/err.log
  python tools/prof_summary.py $O/prof_$m/lstt_kernel_trace.csv 30 > $O/r05d_lstt_isolated_${m}_kernel_stats.md 2>> $O/err.log
  echo "== $m"; cat $O/lstt_$m.json; sed -n 3,4p $O/r05d_lstt_isolated_${m}_kernel_stats.md; sed -n 9,26p $O/r05d_lstt_isolated_${m}_kernel_stats.md | cut -c1-150
done
find $O -name "*.csv" -size +1M -delete
