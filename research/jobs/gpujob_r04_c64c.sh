#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04c64; mkdir -p $O
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -o c64 -- python $OLDPWD/bench.py --config clips64 --batched > $OLDPWD/$O/c64_prof.json 2> $OLDPWD/$O/c64_prof.err )
DB=$(find $O/prof -name "*.db" | head -1)
python - <<PY
import sqlite3
c = sqlite3.connect("$DB").cursor()
rows = list(c.execute("select name, count(*), sum(end-start)/1e3 from kernels group by name order by 3 desc"))
for r in rows[:45]:
    print(r[1], round(r[2]), r[0][:110])
PY
rm -f $O/prof/*.db $O/prof/*/*.db
