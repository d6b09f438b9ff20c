#!/bin/bash
# gate / V projections on a side stream beside the reads (fork / join inside the LSTT graph)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04t; mkdir -p $O
echo -n "lstt isolated fork: "; timeout 300 python tools/lstt_trace.py 2>> $O/lstt.err
echo -n "lstt isolated serial: "; RMEM_FORK=0 timeout 300 python tools/lstt_trace.py 2>> $O/lstt.err
timeout 1500 python -m pytest tests/test_hip_engine.py -q -m gpu -x -k "lstt_forward_vs_oracle or small_clip or closed_loop or paired or prefetch or 480p or graph_caches" > $O/engine_tests.log 2>&1; tail -3 $O/engine_tests.log
for rep in 1 2; do
  echo -n "bench fork: "; timeout 600 python bench.py --no-cpu-baseline --no-dropin 2>> $O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['mean_us'],1))"
  echo -n "bench serial: "; RMEM_FORK=0 timeout 600 python bench.py --no-cpu-baseline --no-dropin 2>> $O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['mean_us'],1))"
done
tail -3 $O/lstt.err
