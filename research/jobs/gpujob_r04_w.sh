#!/bin/bash
# AOT mha_flash: LDS-DMA ring instead of register staging
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04w; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_aot.py -q -m gpu -x > $O/aot_tests.log 2>&1; tail -4 $O/aot_tests.log
timeout 300 python tools/kbench_mha.py > $O/r04w_kbench_mha.json 2> $O/km.err; cat $O/r04w_kbench_mha.json | tr '\n' ' '; echo; tail -2 $O/km.err
for rep in 1 2; do
  echo -n "bench aot: "; timeout 600 python bench.py --model r50_aotl --no-cpu-baseline --no-dropin 2>> $O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['mean_us'], d['roofline']['frac'])"
done
