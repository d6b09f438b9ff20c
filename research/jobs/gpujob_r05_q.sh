#!/bin/bash
# kernel traces of the two other measured shapes: 8 clips per launch, 720p K=8
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r05q; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b8 -o b8 -- python bench.py --batched --clips-per-gpu 8 --steps 40 --warmup 5 --no-cpu-baseline > $O/prof_b8.log 2>&1
python tools/prof_summary.py $O/prof_b8/b8_kernel_trace.csv 30 > $O/r05_bench_batched8_kernel_stats.md 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_720 -o p720 -- python bench.py --config 720p_k8 --gap 2 --steps 40 --warmup 5 --no-cpu-baseline --no-dropin > $O/prof_720.log 2>&1
python tools/prof_summary.py $O/prof_720/p720_kernel_trace.csv 30 > $O/r05_bench_720p_k8_kernel_stats.md 2>&1
find $O -name "*.csv" -size +1M -delete
head -24 $O/r05_bench_batched8_kernel_stats.md; head -24 $O/r05_bench_720p_k8_kernel_stats.md
