#!/bin/bash
# two K splits for the projections under the streaming kernel, last layer through split-K + LayerNorm fold
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04z; mkdir -p $O
echo -n "lstt isolated KS2: "; timeout 300 python tools/lstt_trace.py 2>> $O/lstt.err
echo -n "lstt isolated KS4: "; RMEM_PROJ_KS=4 timeout 300 python tools/lstt_trace.py 2>> $O/lstt.err
timeout 1800 python -m pytest tests/test_hip_engine.py tests/test_hip_batched.py -q -m gpu -x > $O/engine_tests.log 2>&1; tail -3 $O/engine_tests.log
for rep in 1 2; do
  echo -n "bench KS2: "; timeout 600 python bench.py --no-cpu-baseline --no-dropin 2>> $O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['mean_us'],1))"
  echo -n "bench KS4: "; RMEM_PROJ_KS=4 timeout 600 python bench.py --no-cpu-baseline --no-dropin 2>> $O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['mean_us'],1))"
done
