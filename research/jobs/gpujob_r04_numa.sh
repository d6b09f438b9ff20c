#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04numa; mkdir -p $O
{
lscpu | grep -i "numa\|^CPU(s)\|Model name\|Thread\|Socket\|MHz"
echo "--- affinity of this shell"; taskset -p $$; nproc
echo "--- gpu numa"; cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' '; echo
rocm-smi --showtoponuma 2>/dev/null | head -20
cat /sys/fs/cgroup/cpu.max 2>/dev/null
} > $O/topo.txt 2>&1
run() { lab=$1; shift
  echo -n "$lab: "; "$@" timeout 900 python bench.py --config clips64 --batched 2>> $O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), d['config']['rank0_sections_s'])"
}
NN=$(lscpu | awk '/NUMA node\(s\)/{print $3}')
for rep in 1 2 3; do
  run free env
  for n in $(seq 0 $((NN-1))); do
    cpus=$(lscpu | awk -v k="NUMA node$n CPU(s):" 'index($0,k){print $NF}')
    run "node$n($cpus)" taskset -c $cpus
  done
done 2>&1 | tee $O/r04_clips64_batched_numa.txt
