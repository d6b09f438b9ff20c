# round-2 profile of the default bench command: kernel trace + stats, then PMC passes (separate runs, kernel-trace
# only, as MI355X_MICROARCH.md prescribes) for the fused memory-read kernels
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r02_e}
CMD="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- $CMD > gpurun_out/prof_$TAG.log 2>&1
python tools/prof_summary.py gpurun_out/prof_$TAG/${TAG}_kernel_trace.csv 15 > gpurun_out/${TAG}_bench_x3_kernel_stats.md
head -14 gpurun_out/${TAG}_bench_x3_kernel_stats.md
cp gpurun_out/prof_$TAG/${TAG}_kernel_stats.csv gpurun_out/${TAG}_bench_x3_kernel_stats.csv 2>/dev/null
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F16" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_${TAG}_$i -o p -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-dropin > gpurun_out/pmc_${TAG}_$i.log 2>&1
done
TAG=$TAG python - <<'PY'
import csv, glob, json, collections, os
TAG = os.environ["TAG"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/pmc_{TAG}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "read2_kernel" if k.startswith("read2_kernel") else ("read_kernel" if "read_kernel" in k else None)
        if name:
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"note": "rocprofv3 --kernel-trace --pmc <set> in separate passes of python bench.py --steps 6 --warmup 2; mean per dispatch "
               "over every dispatch of the kernel in the run (pre-roll included); FETCH_SIZE / WRITE_SIZE in KB as reported; "
               "hbm_bytes_per_launch = 2*FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE; "
               "SQ_* cycle counters are quad-cycles except SQ_VALU_MFMA_BUSY_CYCLES (cycles, = 32 per 32x32x16 MFMA)"}
for name, cs in agg.items():
    d = {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in cs.items()}
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch"] = (2 * d["FETCH_SIZE"]["mean"] + d["WRITE_SIZE"]["mean"]) * 1024
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CYCLES" in d:
        d["mfma_busy_frac_of_busy_cycles_note"] = "SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles) = matrix-pipe utilisation; see DESIGN.md"
    res[name] = d
if "read2_kernel" in res and "hbm_bytes_per_launch" in res["read2_kernel"]:
    res["hbm_bytes_per_launch"] = res["read2_kernel"]["hbm_bytes_per_launch"]
json.dump(res, open(f"gpurun_out/{TAG}_pmc_read.json", "w"), indent=1)
print(json.dumps(res)[:1500])
PY
