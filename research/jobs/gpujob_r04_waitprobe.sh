#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04cpu; mkdir -p $O
{
for m in default blocking_event devflag4 devflag2 stream_sync poll_sleep; do timeout 120 python research/debug/wait_cpu_probe.py $m 2>&1 | grep -v amdgpu.ids; done
echo "--- env knobs, default mode"
ROC_ACTIVE_WAIT_TIMEOUT=0 timeout 120 python research/debug/wait_cpu_probe.py default 2>&1 | grep -v amdgpu.ids | sed 's/^/ROC_ACTIVE_WAIT_TIMEOUT=0 /'
HSA_ENABLE_INTERRUPT=1 timeout 120 python research/debug/wait_cpu_probe.py default 2>&1 | grep -v amdgpu.ids | sed 's/^/HSA_ENABLE_INTERRUPT=1 /'
AMD_DIRECT_DISPATCH=0 timeout 120 python research/debug/wait_cpu_probe.py default 2>&1 | grep -v amdgpu.ids | sed 's/^/AMD_DIRECT_DISPATCH=0 /'
env | grep -i "^HSA_\|^HIP_\|^ROC\|^AMD_\|^GPU_" 
} | tee $O/r04_wait_cpu_probe.txt
