#!/bin/bash
# same-box A/B: round-4 head (7689e3a, in ab_r04/) against the working tree, alternating
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r05ab; mkdir -p $O
one() { # dir tag args...
  d=$1; tag=$2; shift 2
  (cd $d && python bench.py "$@" 2>/dev/null) | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$tag', ' '.join(sys.argv[1:]), round(d['value'],1))" "$@" | tee -a $O/ab.txt
}
for i in 1 2; do
  one ab_r04 r04 --no-cpu-baseline --no-dropin
  one . r05 --no-cpu-baseline --no-dropin
  one ab_r04 r04 --config clips64
  one . r05 --config clips64
  one ab_r04 r04 --config clips64 --batched
  one . r05 --config clips64 --batched
done
