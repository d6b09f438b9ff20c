cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r05suite; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu --durations=10 --deselect tests/test_driver.py::test_driver_hip_vs_reference_golden > $O/suite.log 2>&1; echo "suite rc $?" >> $O/suite.log
tail -25 $O/suite.log
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps20', d['value'], d['ms_per_step'])"; done
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps100', d['value'], d['ms_per_step'])"; done
