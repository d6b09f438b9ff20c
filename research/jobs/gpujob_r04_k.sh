#!/bin/bash
# XCD-aware dwconv block order, layer-0 LayerNorm straight from the feature map: op tests, kbench, isolated LSTT, engine tests, A/B bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04k; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -q -m gpu -x -k "layernorm or dwconv" > $O/op_tests.log 2>&1; tail -5 $O/op_tests.log
timeout 300 python tools/kbench_gemm.py > $O/r04k_kbench_gemm.json 2> $O/kg.err; grep -E "dwconv|ln2" $O/r04k_kbench_gemm.json
timeout 300 python tools/lstt_trace.py > $O/r04k_lstt_isolated.json 2> $O/lstt_iso.err; cat $O/r04k_lstt_isolated.json; tail -2 $O/lstt_iso.err
timeout 1200 python -m pytest tests/test_hip_engine.py -q -m gpu -x -k "small_clip or closed_loop or prefetch or 480p_teacher_forced or paired" > $O/engine_tests.log 2>&1; tail -5 $O/engine_tests.log
for rep in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --no-dropin > $O/bench_$rep.json 2> $O/err_$rep.log
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04k/bench_*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], round(d["value"], 1), "fps", round(d["ms_per_step"], 3), "ms; read2 in-frame", round(d["roofline"]["mean_us"], 1), "us")
PY
