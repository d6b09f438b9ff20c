#!/bin/bash
# self-read split sweep (isolated LSTT and in the frame); suite subset after the sched memset
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04p; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_engine.py tests/test_hip_batched.py -q -m gpu -x -k "unit_queue or 720p or batched_engine or lstt_many" > $O/tests.log 2>&1; tail -3 $O/tests.log
for ks in 7,2,6 7,2,4 7,2,5 7,2,9 7,2,3; do
  echo -n "$ks isolated: "; RMEM_KS=$ks timeout 300 python tools/lstt_trace.py 2>> $O/lstt.err
done | tee $O/r04p_self_split_sweep.txt
for rep in 1 2; do
  for ks in 7,2,6 7,2,4 7,2,9; do
    echo -n "$ks bench: "; RMEM_KS=$ks timeout 600 python bench.py --no-cpu-baseline --no-dropin 2>> $O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), round(d['ms_per_step'],3))"
  done
done | tee -a $O/r04p_self_split_sweep.txt
