#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03x; mkdir -p $O
timeout 300 python tools/power_probe.py > $O/r03x_power_probe.json 2> $O/power_probe.err
timeout 300 python tools/kbench_read.py > $O/r03x_kbench_read.json 2> $O/kbench_read.err
timeout 300 python tools/kbench.py > $O/r03x_kbench.json 2> $O/kbench.err
timeout 600 python bench.py > $O/r03x_bench_x3.json 2> $O/bench_x3.err
