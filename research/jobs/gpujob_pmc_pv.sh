export TMPDIR=/tmp; mkdir -p gpurun_out
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $set | cut -c1-12 | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_pv_$tag -o p -- python tools/kbench.py --only pv --iters 5 > /dev/null 2>&1
  python - <<PY
import csv, collections, glob
for f in glob.glob("gpurun_out/pmc_pv_$tag/*counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "pv_kernel" in r["Kernel_Name"] or "pv16_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(k, "mean %.3e" % (sum(v)/len(v)), "n", len(v))
PY
done
