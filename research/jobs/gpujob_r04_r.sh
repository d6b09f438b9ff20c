#!/bin/bash
# 128 x 128 items of the streaming projection kernel: tests, kbench both shapes, isolated LSTT, engine tests, A/B in the frame
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04r; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py -q -m gpu -x -k "linear" > $O/op_tests.log 2>&1; tail -4 $O/op_tests.log
for bm in 128 64; do
  RMEM_STREAM_BM=$bm timeout 300 python tools/kbench_gemm.py > $O/r04r_kbench_gemm_bm$bm.json 2>> $O/kg.err; echo "bm $bm: $(cat $O/r04r_kbench_gemm_bm$bm.json | tr '\n' ' ')"
done
RMEM_STREAM_BM=128 timeout 300 python tools/kbench_gemm.py --trace > $O/r04r_stream_trace_bm128.json 2> $O/tr.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r04r/r04r_stream_trace_bm128.json"))
for k, rows in d.items():
    for r in rows:
        c = r["cycles_since_start"]; print(k, r["block"], r["stages"], c[:6], c[-3:])
PY
echo -n "default: "; timeout 300 python tools/lstt_trace.py 2>> $O/lstt.err
echo -n "bm64: "; RMEM_STREAM_BM=64 timeout 300 python tools/lstt_trace.py 2>> $O/lstt.err
timeout 1200 python -m pytest tests/test_hip_engine.py -q -m gpu -x -k "lstt_forward_vs_oracle or small_clip or closed_loop_vs_oracle or 720p" > $O/engine_tests.log 2>&1; tail -3 $O/engine_tests.log
for rep in 1 2; do
  for bm in default 64; do
    echo -n "bench $bm: "; env $( [ $bm != default ] && echo RMEM_STREAM_BM=$bm ) timeout 600 python bench.py --no-cpu-baseline --no-dropin 2>> $O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['mean_us'],1))"
  done
done
