#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04c64d; mkdir -p $O
run() { # label, env...
  lab=$1; shift
  echo -n "$lab: "; env "$@" timeout 900 python bench.py --config clips64 --batched 2>> $O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), d['config']['rank0_sections_s'])"
}
for rep in 1 2; do
  run default A=1
  run tiles RMEM_LINEAR=tiles
  run py_only RMEM_LINEAR_PY=tiles
  run cxx_only RMEM_LINEAR=tiles RMEM_LINEAR_PY=no
done | tee $O/r04_clips64_batched_split.txt
