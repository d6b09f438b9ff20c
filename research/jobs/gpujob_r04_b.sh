#!/bin/bash
# round 4, second job: new tests (RCCL world 1 through bench.py, clips64 two ranks, reduced precision vs the reference's
# autocast fixtures, Swin 9 frames), the bench with the in-frame sampled roofline + its kernel trace, isolated LSTT trace
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04b; mkdir -p $O
timeout 900 python -m pytest tests/test_driver.py -q -m gpu -x -s -k "rccl or world1 or clips64" > $O/driver_new.log 2>&1; tail -5 $O/driver_new.log
timeout 900 python -m pytest tests/test_hip_engine.py -q -m gpu -x -s -k "reduced_precision or test_480p_teacher_forced" > $O/amp.log 2>&1; grep -E "nsplit|autocast|passed|failed" $O/amp.log | cut -c1-600
timeout 1200 python -m pytest tests/test_hip_aot.py -q -m gpu -x -s -k "swin_aot_480x848" > $O/swin.log 2>&1; grep -E "SwinB|passed|failed" $O/swin.log | cut -c1-600
timeout 600 python bench.py > $O/r04b_bench_x3.json 2> $O/bench_x3.err; head -c 2600 $O/r04b_bench_x3.json; tail -3 $O/bench_x3.err
# kernel trace of the same command (no cpu baseline: the trace is about the GPU)
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -o r04b -- python $OLDPWD/bench.py --no-cpu-baseline --no-dropin --steps 60 > $OLDPWD/$O/bench_prof.json 2> $OLDPWD/$O/bench_prof.err )
find $O/prof -name "*kernel_trace.csv" | head -2
KT=$(find $O/prof -name "*kernel_trace.csv" | head -1)
[ -n "$KT" ] && python tools/prof_summary.py $KT 30 > $O/r04b_bench_x3_kernel_stats.md && head -24 $O/r04b_bench_x3_kernel_stats.md
head -c 1500 $O/bench_prof.json
# isolated LSTT: one hipGraph per frame, nothing else on the GPU
timeout 300 python tools/lstt_trace.py > $O/r04b_lstt_isolated.json 2> $O/lstt_iso.err; cat $O/r04b_lstt_isolated.json; tail -2 $O/lstt_iso.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof_lstt -o r04b_lstt -- python $OLDPWD/tools/lstt_trace.py --replays 30 > /dev/null 2> $OLDPWD/$O/lstt_prof.err )
KT=$(find $O/prof_lstt -name "*kernel_trace.csv" | head -1)
[ -n "$KT" ] && python tools/prof_summary.py $KT 25 > $O/r04b_lstt_isolated_kernel_stats.md && cat $O/r04b_lstt_isolated_kernel_stats.md
rm -rf $O/prof/*/*.db $O/prof_lstt/*/*.db 2>/dev/null
du -sh $O
