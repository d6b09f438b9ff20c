#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04e; mkdir -p $O
timeout 300 python tools/kbench_gemm.py --trace > $O/r04e_stream_trace.json 2> $O/tr.err; cat $O/r04e_stream_trace.json; tail -3 $O/tr.err
