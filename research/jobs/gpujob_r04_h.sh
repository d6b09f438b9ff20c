#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04h; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -q -m gpu -x -k "linear" > $O/linear_tests.log 2>&1; tail -5 $O/linear_tests.log
timeout 300 python tools/kbench_gemm.py --trace > $O/r04h_stream_trace.json 2> $O/tr.err; cat $O/r04h_stream_trace.json; tail -3 $O/tr.err
timeout 300 python tools/kbench_gemm.py > $O/r04h_kbench_gemm_stream.json 2> $O/kg.err; cat $O/r04h_kbench_gemm_stream.json; tail -3 $O/kg.err
timeout 300 python tools/lstt_trace.py > $O/r04h_lstt_isolated.json 2> $O/lstt_iso.err; cat $O/r04h_lstt_isolated.json; tail -2 $O/lstt_iso.err
timeout 900 python -m pytest tests/test_hip_engine.py -q -m gpu -x -k "lstt_forward_vs_oracle or small_clip or closed_loop_vs_oracle" > $O/engine_tests.log 2>&1; tail -5 $O/engine_tests.log
