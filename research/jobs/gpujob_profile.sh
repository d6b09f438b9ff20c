# kernel trace + stats of the default bench command (steady state summarised by tools/prof_summary.py)
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r01_e}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_$TAG.log 2>&1
python tools/prof_summary.py gpurun_out/prof_$TAG/${TAG}_kernel_trace.csv 15 > gpurun_out/${TAG}_bench_x3_kernel_stats.md
head -12 gpurun_out/${TAG}_bench_x3_kernel_stats.md
python bench.py 2>/dev/null > gpurun_out/${TAG}_bench_x3.json; cut -c1-400 gpurun_out/${TAG}_bench_x3.json
