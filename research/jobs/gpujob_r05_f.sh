#!/bin/bash
# round 5: the new / changed GPU tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r05f; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_ops.py -q -m gpu -x -k "per_logit or read_bank_logits" -s > $O/logits.log 2>&1; echo "rc $?" >> $O/logits.log
grep -E "max \|HIP|passed|failed|rc " $O/logits.log | tail -12
timeout 2400 python -m pytest tests/test_driver.py -q -m gpu -x -s -k "bench_line_contract or gpus8 or grows_past_ten or batched_drivers_world" > $O/driver.log 2>&1; echo "rc $?" >> $O/driver.log
grep -E "TTA|per_rank|passed|failed|rc |Error|assert" $O/driver.log | tail -12
timeout 1500 python -m pytest tests/test_hip_engine.py tests/test_hip_batched.py -q -m gpu -x -s -k "checkpoint_file or 480p_long or batched_engine_480p or 480p_teacher" > $O/eng.log 2>&1; echo "rc $?" >> $O/eng.log
grep -E "long 480p|off the fp64|frame [0-9]: pixels|batched B=4|vs fp64|passed|failed|rc |Error|assert" $O/eng.log | tail -16
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench.err; tail -c 6000 $O/bench_default.json
