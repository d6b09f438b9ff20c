mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r02_c}
timeout 900 python -m pytest tests/test_hip_engine.py tests/test_hip_ops.py -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; tail -12 gpurun_out/${TAG}_pytest.log
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench_x3.json 2>gpurun_out/${TAG}_bench.err; cut -c1-1500 gpurun_out/${TAG}_bench_x3.json; tail -3 gpurun_out/${TAG}_bench.err
