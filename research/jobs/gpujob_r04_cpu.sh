#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04cpu; mkdir -p $O
run() { lab=$1; shift
  echo -n "$lab: "; env "$@" timeout 900 python bench.py --steps 400 2>> $O/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print(round(d['value'],1), 'issue', round(c['host_issue_ms_per_step'],3), 'cpu', round(c['host_cpu_ms_per_step'],3), c['host_cpu_ms_per_step_by_thread'])"
}
for rep in 1 2; do
  run poll_sleep_wait A=1
  run spin_event RMEM_SPIN_WAIT=1
done | tee $O/r04_host_cpu_poll_wait.txt
timeout 150 python research/debug/world2_probe.py 2 2>&1 | grep "^world" | tee -a $O/r04_host_cpu_poll_wait.txt
