#!/bin/bash
# final kernel trace of the default bench on the final commit (60 timed steps, last 30 frames summarised) + the un-profiled line on the same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04end; mkdir -p $O
timeout 600 python bench.py --no-cpu-baseline --no-dropin > $O/r04end_bench_x3.json 2> $O/bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r04end -- python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-dropin > $O/r04end_bench_x3_under_rocprof.json 2> $O/prof.err
python tools/prof_summary.py $O/prof/r04end_kernel_trace.csv 30 > $O/r04end_bench_x3_kernel_stats.md
cp $O/prof/r04end_kernel_stats.csv $O/r04end_bench_x3_kernel_stats.csv 2>/dev/null
head -30 $O/r04end_bench_x3_kernel_stats.md | cut -c1-170
find $O/prof -name "*.csv" -size +1M -delete
python - <<'PY'
import json
for f in ("r04end_bench_x3.json", "r04end_bench_x3_under_rocprof.json"):
    d = json.loads([l for l in open("gpurun_out/r04end/" + f) if l.startswith("{")][-1])
    print(f, round(d["value"], 1), "fps; read2 by events", round(d["roofline"]["mean_us"], 1), "us over", d["roofline"]["launches"], "launches; isolated", round(d["roofline"]["isolated_mean_us"], 1))
PY
