# AOT path (R50-AOTL + RMem, BASELINE.json configs[0] geometry on the GPU): bench line + kernel stats
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r02_k}
timeout 300 python bench.py --model r50_aotl --no-cpu-baseline --no-dropin > gpurun_out/${TAG}_bench_aot.json 2>/dev/null; cut -c1-1200 gpurun_out/${TAG}_bench_aot.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_aot -o $TAG -- python bench.py --model r50_aotl --steps 20 --warmup 5 --no-cpu-baseline --no-dropin > gpurun_out/prof_${TAG}_aot.log 2>&1
python tools/prof_summary.py gpurun_out/prof_${TAG}_aot/${TAG}_kernel_stats.csv 25 > gpurun_out/${TAG}_bench_aot_kernel_stats.md
head -16 gpurun_out/${TAG}_bench_aot_kernel_stats.md | cut -c1-200
