#!/bin/bash
# kernel-trace summaries of the 720p K=8 and 8-clips-per-launch benches (the default bench's is in the r03f set)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=r03g
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof720 -o t720 -- python bench.py --config 720p_k8 --gap 2 --steps 12 --warmup 4 --no-cpu-baseline --no-dropin > $O/prof720.log 2>&1
python tools/prof_summary.py $O/prof720/t720_kernel_trace.csv 12 > $O/${TAG}_bench_720p_k8_kernel_stats.md
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profb8 -o tb8 -- python bench.py --batched --clips-per-gpu 8 --steps 6 --warmup 2 --no-cpu-baseline > $O/profb8.log 2>&1
python tools/prof_summary.py $O/profb8/tb8_kernel_trace.csv 6 > $O/${TAG}_bench_batched8_kernel_stats.md
head -14 $O/${TAG}_bench_720p_k8_kernel_stats.md | cut -c1-180; head -14 $O/${TAG}_bench_batched8_kernel_stats.md | cut -c1-180
find $O -name "*.csv" -size +1M -delete
