#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04c64t; mkdir -p $O
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/$O/prof -o c64 -- python $OLDPWD/bench.py --config clips64 --batched > $OLDPWD/$O/c64_prof.json 2> $OLDPWD/$O/c64_prof.err )
CSV=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$CSV")))
# last run of clips = timed window: take the last 45% of kernels by time? simply aggregate everything
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg[r["Kernel_Name"][:100]]
    a[0] += 1; a[1] += d
tot = sum(v[1] for v in agg.values())
print("total kernel us", round(tot))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{v[0]:6d} {v[1]:10.0f} us  {v[1]/v[0]:8.1f} us/launch  {k}")
for k, v in agg.items():
    if "bank_" in k or "policy" in k:
        print("POLICY", v[0], round(v[1]), round(v[1] / v[0], 1), k)
PY
rm -rf $O/prof
