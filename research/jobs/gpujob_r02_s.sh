mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r02_s}
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest.log; tail -4 gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py > gpurun_out/${TAG}_bench_x3.json 2>gpurun_out/${TAG}_bench.err; cut -c1-2500 gpurun_out/${TAG}_bench_x3.json; tail -2 gpurun_out/${TAG}_bench.err
timeout 600 python tools/kbench.py > gpurun_out/${TAG}_kbench.json 2>/dev/null; cat gpurun_out/${TAG}_kbench.json | tr -d '\n' | cut -c1-1500; echo
timeout 600 python bench.py --config 720p_k8 --gap 2 --no-cpu-baseline --no-dropin > gpurun_out/${TAG}_bench_720p_k8.json 2>/dev/null; cut -c1-300 gpurun_out/${TAG}_bench_720p_k8.json; echo
timeout 600 python bench.py --clips-per-gpu 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_2clips.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_2clips.json; echo
timeout 600 python bench.py --batched --clips-per-gpu 8 --steps 40 > gpurun_out/${TAG}_bench_batched8.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_batched8.json; echo; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_batched8.json')); print(d.get('roofline'))"
timeout 600 python bench.py --batched --clips-per-gpu 4 --steps 40 > gpurun_out/${TAG}_bench_batched4.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_batched4.json; echo
timeout 600 python bench.py --model r50_aotl --no-cpu-baseline --no-dropin > gpurun_out/${TAG}_bench_aot.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_aot.json; echo
