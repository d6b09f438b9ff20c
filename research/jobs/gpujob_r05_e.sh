#!/bin/bash
# in-frame A/B of the projection paths: default bench, alternating, two repeats
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r05e; mkdir -p $O
for i in 1 2; do
  for m in 0 planes fused; do
    RMEM_ROWRES=$m timeout 600 python bench.py --no-cpu-baseline --no-dropin > $O/bench_${m}_$i.json 2>> $O/err.log
    python - <<PY
import json
d = json.loads([l for l in open("$O/bench_${m}_$i.json") if l.startswith("{")][-1])
print("$m", $i, round(d["value"], 1), "fps", round(d["ms_per_step"], 3), "ms; read2", round(d["roofline"]["mean_us"], 1), "us; mism", d.get("parity", {}).get("mask_mismatch_px"))
PY
  done
done
