mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r02_n}
timeout 900 python -m pytest tests/test_hip_batched.py -x -q -s > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest.log; tail -25 gpurun_out/${TAG}_pytest.log
for B in 2 4 8; do
timeout 600 python bench.py --batched --clips-per-gpu $B --steps 40 > gpurun_out/${TAG}_bench_b$B.json 2>gpurun_out/${TAG}_bench_b$B.err; cut -c1-1500 gpurun_out/${TAG}_bench_b$B.json; tail -3 gpurun_out/${TAG}_bench_b$B.err
done
