#!/bin/bash
# round 4: where the small GEMMs' time goes (tools/kbench_gemm.py), kbench with graph-replay timing
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04c; mkdir -p $O
timeout 600 python tools/kbench_gemm.py > $O/r04c_kbench_gemm.json 2> $O/kbench_gemm.err; cat $O/r04c_kbench_gemm.json; tail -3 $O/kbench_gemm.err
timeout 600 python tools/kbench.py > $O/r04c_kbench.json 2> $O/kbench.err; cat $O/r04c_kbench.json; tail -3 $O/kbench.err
