# verify HEAD on the GPU box: -m gpu tests, headline bench, the other configs, per-kernel micro-benchmark
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r01_g}
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest.log
tail -4 gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py 2>gpurun_out/${TAG}_bench.err > gpurun_out/${TAG}_bench_x3.json; cut -c1-300 gpurun_out/${TAG}_bench_x3.json
timeout 300 python bench.py --no-cpu-baseline --model r50_aotl 2>/dev/null > gpurun_out/${TAG}_bench_aot.json; cut -c1-200 gpurun_out/${TAG}_bench_aot.json
timeout 300 python tools/kbench.py > gpurun_out/${TAG}_kbench.json 2>gpurun_out/${TAG}_kbench.err; cat gpurun_out/${TAG}_kbench.json
