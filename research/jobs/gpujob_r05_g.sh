#!/bin/bash
# 8 ranks on one device vs one rank with the same eight clips: which clip hashes differ, and is the one-rank run repeatable
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r05g; mkdir -p $O
export RMEM_DEVICE_OVERRIDE=0 RMEM_DIST_BACKEND=gloo
python bench.py --gpus 8 --config clips64 --clips-per-rank 1 --clip-frames 4 > $O/w8.json 2> $O/w8.err
python bench.py --gpus 1 --config clips64 --clips-per-rank 8 --clip-frames 4 > $O/w1a.json 2> $O/w1a.err
python bench.py --gpus 1 --config clips64 --clips-per-rank 8 --clip-frames 4 > $O/w1b.json 2> $O/w1b.err
python bench.py --gpus 2 --config clips64 --clips-per-rank 4 --clip-frames 4 > $O/w2.json 2> $O/w2.err
python bench.py --gpus 8 --config clips64 --clips-per-rank 1 --clip-frames 4 > $O/w8b.json 2> $O/w8b.err
python - <<'PY'
import json
for f in ("w8", "w8b", "w1a", "w1b", "w2"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r05g/{f}.json") if l.startswith("{")][-1])
        print(f, [h[:6] for h in d["clip_sha256"]], round(d["value"], 1))
    except Exception as e:
        print(f, "failed", e)
PY
