cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, final tree: smoke, the driver's bench command, the GPU suite
O=$PWD/gpurun_out/r06final; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_driver_cmd.json 2> $O/bench_driver.err; python -c "
import json; d=json.load(open('$O/r06_bench_driver_cmd.json')); print('driver cmd', round(d['value'],1), d['steps'], d['warmup'], d['roofline']['frac'], d['cpu_baseline']['value'])"
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -12 | tee $O/pytest_gpu.log
