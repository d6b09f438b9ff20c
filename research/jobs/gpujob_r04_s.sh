#!/bin/bash
# combine kernels: partial loads hoisted in front of the statistics
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04s; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py -q -m gpu -x -k "read_bank or read_window" > $O/op_tests.log 2>&1; tail -3 $O/op_tests.log
timeout 600 python tools/split_sweep.py default > $O/r04s_combine.txt 2> $O/sw.err; cat $O/r04s_combine.txt
echo -n "lstt isolated: "; timeout 300 python tools/lstt_trace.py 2>> $O/lstt.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof_lstt -o r04s_lstt -- python $OLDPWD/tools/lstt_trace.py --replays 30 > /dev/null 2> $OLDPWD/$O/lstt_prof.err )
DB=$(find $O/prof_lstt -name "*.db" | head -1)
[ -n "$DB" ] && python tools/prof_summary.py $DB 25 > $O/r04s_lstt_isolated_kernel_stats.md && sed -n 9,26p $O/r04s_lstt_isolated_kernel_stats.md | cut -c1-130
rm -f $O/prof_lstt/*.db $O/prof_lstt/*/*.db
timeout 1200 python -m pytest tests/test_hip_engine.py tests/test_hip_batched.py -q -m gpu -x -k "lstt_forward_vs_oracle or small_clip or closed_loop_vs_oracle or paired or lstt_many or long_clip" > $O/engine_tests.log 2>&1; tail -3 $O/engine_tests.log
for rep in 1 2; do
  echo -n "bench: "; timeout 600 python bench.py --no-cpu-baseline --no-dropin 2>> $O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['mean_us'],1))"
done
