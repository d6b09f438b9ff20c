mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r01_f_aot}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python bench.py --model r50_aotl --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_$TAG.log 2>&1
python - <<PY
import csv, sys
sys.argv = ["x", "gpurun_out/prof_$TAG/${TAG}_kernel_trace.csv", "15"]
sys.path.insert(0, "tools")
import prof_summary as ps
# AOT has no gn2_apply marker: use the once-per-frame id_assign kernel
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("id_assign_kernel")]
frames = 15
seg = rows[marks[-frames - 1] + 1: marks[-1] + 1]
agg = {}
for r in seg:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(r["Kernel_Name"], [0, 0.0]); a[0] += 1; a[1] += d
tot = sum(v[1] for v in agg.values())
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3 / frames
print(f"AOT: kernel time {tot/frames:.1f} us/frame, wall span {span:.1f} us/frame")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{v[1]/frames:8.1f} us/frame  {v[0]/frames:5.1f} calls  {v[1]/v[0]:7.2f} avg  {k[:90]}")
PY
