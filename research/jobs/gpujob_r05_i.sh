#!/bin/bash
# the other bench lines with their parity / cpu_baseline blocks (AOT block, Swin-B backbone, 720p K=8)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r05i; mkdir -p $O
timeout 900 python bench.py --model r50_aotl --no-dropin > $O/r05i_bench_aot.json 2> $O/aot.err
timeout 900 python bench.py --model swinb_aotl --no-dropin --cpu-frames 2 > $O/r05i_bench_swin.json 2> $O/swin.err
timeout 1500 python bench.py --config 720p_k8 --gap 2 --no-dropin --steps 40 > $O/r05i_bench_720p_k8.json 2> $O/720.err
for f in aot swin 720p_k8; do python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/r05i_bench_$f.json") if l.startswith("{")][-1])
    print("$f", round(d["value"], 1), "fps; roofline", d["roofline"]["kernel"][:40], round(d["roofline"]["frac"], 3), "; cpu", d.get("cpu_baseline"), "; mism", d.get("mask_mismatch_px"), "iou", d.get("iou_vs_oracle"), d.get("iou_ids"), "evict ok", d.get("eviction_sequence_equal"))
except Exception as e:
    print("$f FAILED", e); print(open("$O/" + {"aot": "aot", "swin": "swin", "720p_k8": "720"}["$f"] + ".err").read()[-1500:])
PY
done
