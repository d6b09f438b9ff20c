mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r02_f}
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest.log; tail -6 gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py > gpurun_out/${TAG}_bench_x3.json 2>gpurun_out/${TAG}_bench.err; cut -c1-2600 gpurun_out/${TAG}_bench_x3.json; tail -2 gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --config clips64 --clips-per-rank 2 > gpurun_out/${TAG}_bench_clips64_2.json 2>gpurun_out/${TAG}_clips64.err; cut -c1-900 gpurun_out/${TAG}_bench_clips64_2.json; tail -2 gpurun_out/${TAG}_clips64.err
