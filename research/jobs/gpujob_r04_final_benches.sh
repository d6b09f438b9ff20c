#!/bin/bash
# the bench lines of every configuration on the final kernels (one box)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-r04g}
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python bench.py --no-cpu-baseline > $O/${TAG}_bench_x3.json 2> $O/bench_x3.err
timeout 600 python bench.py --config 720p_k8 --gap 2 --no-cpu-baseline --no-dropin > $O/${TAG}_bench_720p_k8.json 2> $O/bench_720.err
timeout 600 python bench.py --batched --clips-per-gpu 8 --no-cpu-baseline > $O/${TAG}_bench_batched8.json 2> $O/bench_b8.err
timeout 600 python bench.py --batched --clips-per-gpu 4 --no-cpu-baseline > $O/${TAG}_bench_batched4.json 2> $O/bench_b4.err
timeout 600 python bench.py --clips-per-gpu 2 --no-cpu-baseline --no-dropin > $O/${TAG}_bench_2clips.json 2> $O/bench_2c.err
timeout 600 python bench.py --model r50_aotl --no-cpu-baseline --no-dropin > $O/${TAG}_bench_aot.json 2> $O/bench_aot.err
timeout 600 python bench.py --model swinb_aotl --no-cpu-baseline --no-dropin > $O/${TAG}_bench_swin.json 2> $O/bench_swin.err
timeout 900 python bench.py --config clips64 > $O/${TAG}_bench_clips64.json 2> $O/bench_c64.err
timeout 900 python bench.py --config clips64 --batched > $O/${TAG}_bench_clips64_batched.json 2> $O/bench_c64b.err
timeout 300 python tools/lstt_trace.py > $O/${TAG}_lstt_isolated.json 2> $O/lstt.err
timeout 300 python tools/lstt_trace.py --h 46 --w 81 --cap 8 > $O/${TAG}_lstt_isolated_720p.json 2>> $O/lstt.err
for f in $O/${TAG}_*.json; do python -c "
import json,sys
try:
    d=json.load(open('$f')); r=d.get('roofline') or {}; print('$f'.split('/')[-1], round(d.get('value',0),1) if 'value' in d else d, r.get('frac'), r.get('mean_us'))
except Exception as e: print('$f', 'ERR', e)
"; done
