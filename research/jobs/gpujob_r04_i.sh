#!/bin/bash
# A/B in the frame: streaming projection kernel vs the tile kernels (same box, alternating)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04i; mkdir -p $O
for rep in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --no-dropin > $O/bench_stream_$rep.json 2> $O/err_s$rep.log
  RMEM_LINEAR=tiles timeout 600 python bench.py --no-cpu-baseline --no-dropin > $O/bench_tiles_$rep.json 2> $O/err_t$rep.log
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04i/bench_*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], round(d["value"], 1), "fps", round(d["ms_per_step"], 3), "ms; read2 in-frame", round(d["roofline"]["mean_us"], 1), "us; host cpu", round(d["config"]["host_cpu_ms_per_step"], 2))
PY
