#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04c64e; mkdir -p $O
run() { lab=$1; shift
  echo -n "$lab: "; env "$@" timeout 900 python bench.py --config clips64 2>> $O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), d['config']['rank0_sections_s'])"
}
for rep in 1 2 3; do
  run poll_wait A=1
  run spin_wait RMEM_SPIN_WAIT=1
done | tee $O/r04_clips64_wait.txt
