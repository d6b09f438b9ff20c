#!/bin/bash
# final round-3 evidence run: whole GPU suite, then the profile set (research/jobs/gpujob_profile_r03.sh), kernel micro-benches
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03f; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
bash research/jobs/gpujob_profile_r03.sh r03f > $O/profile_job.log 2>&1
timeout 300 python tools/kbench.py > $O/r03f_kbench.json 2> $O/kbench.err
timeout 300 python tools/kbench_read.py > $O/r03f_kbench_read.json 2> $O/kbench_read.err
timeout 600 python bench.py --batched --clips-per-gpu 2 --no-cpu-baseline > $O/r03f_bench_batched2.json 2> $O/bench_b2.err
timeout 600 python bench.py --model r50_aotl --no-cpu-baseline --no-dropin > $O/r03f_bench_aot.json 2> $O/bench_aot.err
RMEM_DIST_BACKEND=gloo RMEM_DEVICE_OVERRIDE=0 timeout 600 python bench.py --gpus 2 --steps 10 --no-cpu-baseline --no-dropin > $O/r03f_bench_gpus2_one_device.json 2> $O/bench_g2.err
tail -2 $O/profile_job.log
timeout 300 python tools/power_probe.py > $O/r03f_power_probe.json 2> $O/power_probe.err
