mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r02_r}
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest.log; tail -6 gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py > gpurun_out/${TAG}_bench_x3.json 2>gpurun_out/${TAG}_bench.err; cut -c1-400 gpurun_out/${TAG}_bench_x3.json; tail -2 gpurun_out/${TAG}_bench.err
for ks in "6,3,9" "6,3,4" "7,2,6"; do RMEM_KS=$ks timeout 600 python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ks $ks fps %.1f read2 iso %.1f' % (d['value'], d['roofline']['isolated_mean_us']))"; done
timeout 600 python tools/kbench.py > gpurun_out/${TAG}_kbench.json 2>/dev/null; cat gpurun_out/${TAG}_kbench.json | tr -d '\n' | cut -c1-900; echo
