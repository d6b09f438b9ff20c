mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r02_q}
timeout 900 python -m pytest tests/test_hip_batched.py -x -q -s > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest.log; tail -8 gpurun_out/${TAG}_pytest.log
timeout 900 python bench.py --config clips64 --clips-per-rank 8 --batched > gpurun_out/${TAG}_bench_clips64_batched.json 2>gpurun_out/${TAG}_c64b.err; cut -c1-1200 gpurun_out/${TAG}_bench_clips64_batched.json; tail -2 gpurun_out/${TAG}_c64b.err
timeout 900 python bench.py --config clips64 --clips-per-rank 8 > gpurun_out/${TAG}_bench_clips64.json 2>gpurun_out/${TAG}_c64.err; cut -c1-1200 gpurun_out/${TAG}_bench_clips64.json; tail -2 gpurun_out/${TAG}_c64.err
timeout 600 python bench.py --batched --clips-per-gpu 8 --steps 40 > gpurun_out/${TAG}_bench_batched8.json 2>/dev/null; cut -c1-1500 gpurun_out/${TAG}_bench_batched8.json
timeout 600 python bench.py --batched --clips-per-gpu 4 --steps 40 > gpurun_out/${TAG}_bench_batched4.json 2>/dev/null; cut -c1-300 gpurun_out/${TAG}_bench_batched4.json
CMD="python bench.py --batched --clips-per-gpu 8 --steps 20 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- $CMD > gpurun_out/prof_$TAG.log 2>&1
python tools/prof_summary.py gpurun_out/prof_$TAG/${TAG}_kernel_trace.csv 15 > gpurun_out/${TAG}_batched8_kernel_stats.md
head -30 gpurun_out/${TAG}_batched8_kernel_stats.md | cut -c1-160
rm -f gpurun_out/prof_$TAG/*kernel_trace.csv
