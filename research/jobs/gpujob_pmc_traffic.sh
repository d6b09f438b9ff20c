# HBM-side traffic of the dominant kernel (long-term pv_kernel<3>) from rocprofv3 PMC counters, collected in
# separate passes of the bench command as MI355X_MICROARCH.md prescribes (no trace domains besides kernel-trace)
export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r01_h}
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_$c -o p -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_$c.log 2>&1
done
python - <<PY
import csv, glob, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "pv16_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                vals.append(float(r["Counter_Value"]))
    vals.sort()
    # the long-term launches (T=4 bank: 3 of the 9 P.V launches of a frame) are the largest third
    top = vals[-max(1, len(vals) // 3):]
    out[c] = {"launches_all": len(vals), "launches_long": len(top), "mean_KB_long": sum(top) / len(top), "max_KB": vals[-1] if vals else 0}
fetch = 2 * out["FETCH_SIZE"]["mean_KB_long"] * 1024      # gfx950: FETCH_SIZE reports half of wide coalesced reads
write = out["WRITE_SIZE"]["mean_KB_long"] * 1024
res = {"note": "rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate runs of python bench.py --steps 6 --warmup 2); KB as reported; FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md HBM section); long-term launches = the largest third of the pv16_kernel dispatches (bank reads: long-term and self)",
       "pv16_kernel long-term (480p K=4, T=4)": out, "hbm_bytes_per_launch": fetch + write,
       "algorithmic_bytes_per_launch": 84410368}
json.dump(res, open(f"gpurun_out/${TAG}_pmc_pv_long.json", "w"), indent=1)
print(json.dumps(res)[:600])
PY
