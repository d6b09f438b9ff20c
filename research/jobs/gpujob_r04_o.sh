#!/bin/bash
# XCD-local item order of the streaming projection kernel: tests, kbench, trace, isolated LSTT, A/B in the frame
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04o; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -q -m gpu -x -k "linear or groupnorm or layernorm" > $O/op_tests.log 2>&1; tail -3 $O/op_tests.log
timeout 300 python tools/kbench_gemm.py > $O/r04o_kbench_gemm_xcd.json 2> $O/kg.err; cat $O/r04o_kbench_gemm_xcd.json | tr '\n' ' '; echo
RMEM_STREAM_ORDER=plain timeout 300 python tools/kbench_gemm.py > $O/r04o_kbench_gemm_plain.json 2>> $O/kg.err; cat $O/r04o_kbench_gemm_plain.json | tr '\n' ' '; echo
timeout 300 python tools/kbench_gemm.py --trace > $O/r04o_stream_trace.json 2> $O/tr.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r04o/r04o_stream_trace.json"))
for k, rows in d.items():
    for r in rows:
        c = r["cycles_since_start"]; print(k, r["block"], r["stages"], c[0], c[1], c[-2], c[-1])
PY
timeout 300 python tools/lstt_trace.py 2> $O/lstt.err | tee $O/r04o_lstt_isolated_xcd.json
RMEM_STREAM_ORDER=plain timeout 300 python tools/lstt_trace.py 2>> $O/lstt.err | tee $O/r04o_lstt_isolated_plain.json
for rep in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --no-dropin > $O/bench_xcd_$rep.json 2> $O/err.log
  RMEM_STREAM_ORDER=plain timeout 600 python bench.py --no-cpu-baseline --no-dropin > $O/bench_plain_$rep.json 2>> $O/err.log
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04o/bench_*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], round(d["value"], 1), "fps", round(d["ms_per_step"], 3), "ms; read2 in-frame", round(d["roofline"]["mean_us"], 1))
PY
