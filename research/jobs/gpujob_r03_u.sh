#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03u; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_aot.py tests/test_hip_batched.py tests/test_hip_ops.py -q -m gpu -k "mha or aot or ragged or batched_clip_driver or swin" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python tools/kbench_mha.py > $O/r03u_kbench_mha.json 2> $O/kbench_mha.err; tail -c 600 $O/r03u_kbench_mha.json
timeout 600 python bench.py --model r50_aotl --no-cpu-baseline --no-dropin > $O/r03u_bench_aot.json 2> $O/bench_aot.err
timeout 600 python bench.py --batched --clips-per-gpu 4 --no-cpu-baseline > $O/r03u_bench_batched4.json 2> $O/bench_b4.err
RMEM_DIST_BACKEND=gloo RMEM_DEVICE_OVERRIDE=0 timeout 600 python bench.py --gpus 2 --steps 10 --no-cpu-baseline --no-dropin > $O/r03u_bench_gpus2_one_device.json 2> $O/bench_g2.err
source <(sed -n '/^SETS=/,/^}/p' research/jobs/gpujob_profile_r03.sh)
TAG=r03u
pmc aot python bench.py --model r50_aotl --steps 6 --warmup 2 --no-cpu-baseline --no-dropin
