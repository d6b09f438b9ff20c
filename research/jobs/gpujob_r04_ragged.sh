#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04ragged; mkdir -p $O
for mode in "--batched" ""; do
  timeout 600 python bench.py --config clips64 --ragged $mode > $O/r04_bench_ragged${mode/--/_}.json 2> $O/err${mode/--/_}.log
  python - <<PY
import json
d=json.loads(open("$O/r04_bench_ragged${mode/--/_}.json").readline())
print("$mode", round(d["value"],1), d["config"]["workload"], d["config"]["queue_stats_rank0"], d["config"]["per_rank_seconds"])
PY
done
RMEM_FORCE_DIST=1 timeout 600 python bench.py --config clips64 --ragged --batched 2> $O/err_rccl.log | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('rccl world 1', round(d['value'],1), d['config']['dist_backend'])"
tail -3 $O/err_batched.log
