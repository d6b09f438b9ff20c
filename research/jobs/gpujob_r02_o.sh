mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r02_o}
timeout 900 python -m pytest tests/test_hip_batched.py -x -q -s > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest.log; tail -8 gpurun_out/${TAG}_pytest.log
CMD="python bench.py --batched --clips-per-gpu 4 --steps 20 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- $CMD > gpurun_out/prof_$TAG.log 2>&1
python tools/prof_summary.py gpurun_out/prof_$TAG/${TAG}_kernel_trace.csv 20 > gpurun_out/${TAG}_batched4_kernel_stats.md
head -60 gpurun_out/${TAG}_batched4_kernel_stats.md
