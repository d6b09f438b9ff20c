#!/bin/bash
# round 3, GPU job N: additive LDS layouts of the read kernel; batched benches without the MIOpen switch
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03n; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_ops.py -q -m gpu -k "read_" > $O/pytest_read.log 2>&1; echo "pytest rc $?" >> $O/pytest_read.log
timeout 300 python tools/kbench_read.py > $O/kbench_read.json 2> $O/kbench_read.err
timeout 300 python tools/kbench.py > $O/kbench.json 2> $O/kbench.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_1 -o p -- python tools/kbench_read.py --only long --splits 9 --iters 5 --no-trace > $O/pmc_1.log 2>&1
python - <<'PY' > gpurun_out/r03n/pmc_summary.txt 2>&1
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r03n/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("read64"):
            agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k, {c: round(sum(v) / len(v) / 1e6, 3) for c, v in cs.items()}, "n", len(next(iter(cs.values()))))
PY
find $O -name "*.csv" -size +1M -delete
timeout 600 python bench.py --batched --clips-per-gpu 8 --no-cpu-baseline > $O/bench_batched8.json 2> $O/bench_b8.err
timeout 600 python bench.py --batched --clips-per-gpu 4 --no-cpu-baseline > $O/bench_batched4.json 2> $O/bench_b4.err
timeout 600 python bench.py --no-cpu-baseline --no-dropin > $O/bench_x3.json 2> $O/bench_x3.err
tail -3 $O/pytest_read.log; cat $O/pmc_summary.txt; for f in batched8 batched4 x3; do python -c "
import json; d=json.load(open('$O/bench_$f.json')); print('$f', round(d['value'],1), d['roofline'].get('mean_us'), d['roofline'].get('frac'))"; done
