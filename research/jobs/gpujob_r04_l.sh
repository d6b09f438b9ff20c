#!/bin/bash
# A/B on one box: every switch of the round against the default (alternating, two repeats)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04l; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -q -m gpu -x -k "layernorm or dwconv" > $O/op_tests.log 2>&1; tail -3 $O/op_tests.log
for rep in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --no-dropin > $O/bench_default_$rep.json 2> $O/err.log
  RMEM_DW_ORDER=grid timeout 600 python bench.py --no-cpu-baseline --no-dropin > $O/bench_dwgrid_$rep.json 2>> $O/err.log
  RMEM_LN_CN=0 timeout 600 python bench.py --no-cpu-baseline --no-dropin > $O/bench_lncn0_$rep.json 2>> $O/err.log
  RMEM_LINEAR=tiles timeout 600 python bench.py --no-cpu-baseline --no-dropin > $O/bench_tiles_$rep.json 2>> $O/err.log
  RMEM_LINEAR=tiles RMEM_LN_CN=0 RMEM_DW_ORDER=grid timeout 600 python bench.py --no-cpu-baseline --no-dropin > $O/bench_alloff_$rep.json 2>> $O/err.log
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04l/bench_*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], round(d["value"], 1), "fps", round(d["ms_per_step"], 3), "ms; read2 in-frame", round(d["roofline"]["mean_us"], 1), "us")
PY
