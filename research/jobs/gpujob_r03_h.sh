#!/bin/bash
# round 3, GPU job H: read64 kernel -- V-latency experiment, PMC counters of the isolated long-term read, encoder race probe
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03h; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/kbench_read.py --old > $O/kbench_read.json 2> $O/kbench_read.err
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$i -o p -- python tools/kbench_read.py --only long --splits 9 --iters 5 --no-trace > $O/pmc_$i.log 2>&1
done
python - <<'PY' > gpurun_out/r03h/pmc_summary.txt 2>&1
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r03h/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "read" in k:
            agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k, {c: round(sum(v) / len(v) / 1e6, 3) for c, v in cs.items()}, "n", len(next(iter(cs.values()))))
PY
find $O -name "*.csv" -size +2M -delete
timeout 600 python tools/encoder_race_probe.py > $O/encoder_race_probe.json 2> $O/encoder_race_probe.err
cat $O/pmc_summary.txt
