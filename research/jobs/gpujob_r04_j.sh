#!/bin/bash
# kernel trace of the default bench with the streaming projection kernel; whole GPU suite with the new silu
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04j; mkdir -p $O
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -o r04j -- python $OLDPWD/bench.py --no-cpu-baseline --no-dropin --steps 60 > $OLDPWD/$O/bench_prof.json 2> $OLDPWD/$O/bench_prof.err )
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/prof_summary.py $DB 30 > $O/r04j_bench_x3_kernel_stats.md && head -34 $O/r04j_bench_x3_kernel_stats.md | cut -c1-150
rm -f $O/prof/*.db $O/prof/*/*.db
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof_lstt -o r04j_lstt -- python $OLDPWD/tools/lstt_trace.py --replays 30 > /dev/null 2> $OLDPWD/$O/lstt_prof.err )
DB=$(find $O/prof_lstt -name "*.db" | head -1)
[ -n "$DB" ] && python tools/prof_summary.py $DB 25 > $O/r04j_lstt_isolated_kernel_stats.md && sed -n 1,24p $O/r04j_lstt_isolated_kernel_stats.md | cut -c1-150
rm -f $O/prof_lstt/*.db $O/prof_lstt/*/*.db
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
