#!/bin/bash
# after the unit queue / block order / 720p split changes: the configurations they touch, re-measured (tag r03g)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=r03g
O=gpurun_out/$TAG; mkdir -p $O
source <(sed -n '/^SETS=/,/^}/p' research/jobs/gpujob_profile_r03.sh)
pmc 720p_k8 python bench.py --config 720p_k8 --gap 2 --steps 6 --warmup 2 --no-cpu-baseline --no-dropin
pmc batched8 python bench.py --batched --clips-per-gpu 8 --steps 4 --warmup 1 --no-cpu-baseline
timeout 600 python bench.py --config 720p_k8 --gap 2 --no-cpu-baseline --no-dropin > $O/${TAG}_bench_720p_k8.json 2> $O/bench_720.err
timeout 600 python bench.py --batched --clips-per-gpu 8 --no-cpu-baseline > $O/${TAG}_bench_batched8.json 2> $O/bench_b8.err
timeout 600 python bench.py --batched --clips-per-gpu 2 --no-cpu-baseline > $O/${TAG}_bench_batched2.json 2> $O/bench_b2.err
timeout 900 python bench.py --config clips64 --batched > $O/${TAG}_bench_clips64_batched.json 2> $O/bench_c64b.err
timeout 600 python bench.py > $O/${TAG}_bench_x3.json 2> $O/bench_x3.err
