#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04devpol; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_batched.py -x -q -m gpu -s 2>&1 | tail -40 > $O/batched_tests.log
tail -22 $O/batched_tests.log | cut -c1-300
run() { lab=$1; shift
  echo -n "$lab: "; env "$@" timeout 900 python bench.py --config clips64 --batched 2>> $O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), d['config']['rank0_sections_s'], d['clip_sha256'][:2])"
}
for rep in 1 2 3; do
  run device_policy A=1
  run host_policy RMEM_HOST_POLICY=1
done | tee $O/r04_clips64_batched_policy.txt
for rep in 1 2; do
echo -n "batched8 steady device: "; timeout 600 python bench.py --batched --clips-per-gpu 8 2>>$O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],1))"
echo -n "batched8 steady host: "; RMEM_HOST_POLICY=1 timeout 600 python bench.py --batched --clips-per-gpu 8 2>>$O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],1))"
done | tee -a $O/r04_clips64_batched_policy.txt
