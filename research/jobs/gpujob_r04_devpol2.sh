#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04devpol; mkdir -p $O
run() { lab=$1; shift
  echo -n "$lab: "; env "$@" timeout 900 python bench.py --config clips64 $BATCHED 2>> $O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), d['config']['rank0_sections_s'], d['clip_sha256'][:2])"
}
for rep in 1 2 3; do
  BATCHED=--batched run batched_device_policy A=1
  BATCHED=--batched run batched_host_policy RMEM_HOST_POLICY=1
  BATCHED= run per_clip A=1
done | tee $O/r04_clips64_pinned.txt
