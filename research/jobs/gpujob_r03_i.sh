#!/bin/bash
# round 3, GPU job I: full -m gpu suite on the read64 kernel, per-kernel bench with split sweeps, headline bench,
# encoder race probe at the small geometry
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03i; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --durations=10 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
timeout 300 python tools/kbench.py > $O/kbench.json 2> $O/kbench.err
RMEM_KS=6,3,6 timeout 300 python tools/kbench.py --only reads > $O/kbench_636.json 2>> $O/kbench.err
RMEM_KS=8,1,9 timeout 300 python tools/kbench.py --only reads > $O/kbench_819.json 2>> $O/kbench.err
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
PH=97 PW=129 timeout 600 python tools/encoder_race_probe.py > $O/encoder_race_probe_97x129.json 2> $O/encoder_race_probe.err
tail -5 $O/pytest_gpu.log; cat $O/bench.json | head -c 1500
