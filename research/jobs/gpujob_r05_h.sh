#!/bin/bash
# round 5: lane-swap / DPP reductions everywhere (no ds_bpermute): parity of every kernel that reduces across lanes, then A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r05h; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_ops.py tests/test_hip_aot.py -q -m gpu -x > $O/ops.log 2>&1; echo "rc $?" >> $O/ops.log; tail -3 $O/ops.log
timeout 1500 python -m pytest tests/test_hip_engine.py -q -m gpu -x -k "lstt_forward or small_clip or 480p_long or 720p or paired or unit_queue" > $O/eng.log 2>&1; echo "rc $?" >> $O/eng.log; tail -3 $O/eng.log
timeout 900 python -m pytest tests/test_hip_batched.py -q -m gpu -x -s -k "every_slot" > $O/slots.log 2>&1; echo "rc $?" >> $O/slots.log; grep -E "8 slots|off the fp64|passed|failed|rc " $O/slots.log | tail -5
for i in 1 2 3; do timeout 300 python tools/lstt_trace.py >> $O/lstt_iso.txt 2>> $O/err.log; done; cat $O/lstt_iso.txt
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-dropin > $O/bench_$i.json 2>> $O/err.log; python - <<PY
import json
d = json.loads([l for l in open("$O/bench_$i.json") if l.startswith("{")][-1])
print("bench", $i, round(d["value"], 1), "fps; read2", round(d["roofline"]["mean_us"], 1), "us;", [(k["kernel"][:14], round(k["us_per_frame"])) for k in d["roofline"]["kernels"]])
PY
done
timeout 600 python bench.py --config clips64 --batched > $O/clips64_batched.json 2>> $O/err.log; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05h/clips64_batched.json") if l.startswith("{")][-1])
print("clips64 batched N=1:", round(d["value"], 1), "frames/s", d["config"]["rank0_sections_s"])
PY
