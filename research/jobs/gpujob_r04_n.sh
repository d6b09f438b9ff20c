#!/bin/bash
# uneven long-term key splits: tests, isolated sweep, A/B in the frame
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04n; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py -q -m gpu -x -k "read_bank or read_window" > $O/read_tests.log 2>&1; tail -3 $O/read_tests.log
timeout 900 python tools/split_sweep.py default 7,2,6 8,2,6,7,15 9,2,6,7,14 10,2,6,7,14 9,2,6,8,13 9,3,6,7,14 8,3,6,7,15 > $O/r04n_split_sweep.txt 2> $O/sweep.err; cat $O/r04n_split_sweep.txt; tail -2 $O/sweep.err
timeout 600 python -m pytest tests/test_hip_engine.py -q -m gpu -x -k "paired_launches or unit_queue or small_clip" > $O/engine_tests.log 2>&1; tail -3 $O/engine_tests.log
for rep in 1 2; do
  RMEM_UNEVEN=0 timeout 600 python bench.py --no-cpu-baseline --no-dropin > $O/bench_even_$rep.json 2> $O/err.log
  RMEM_KS=8,2,6,7,15 timeout 600 python bench.py --no-cpu-baseline --no-dropin > $O/bench_u8_$rep.json 2>> $O/err.log
  RMEM_KS=9,2,6,7,14 timeout 600 python bench.py --no-cpu-baseline --no-dropin > $O/bench_u9_$rep.json 2>> $O/err.log
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04n/bench_*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], round(d["value"], 1), "fps", round(d["ms_per_step"], 3), "ms; read2 in-frame", round(d["roofline"]["mean_us"], 1), "us iso", round(d["roofline"]["isolated_mean_us"], 1))
PY
