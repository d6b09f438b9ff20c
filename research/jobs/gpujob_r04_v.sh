#!/bin/bash
# dwconv: register budget for 5 / 6 / 8 waves per SIMD (spills) against the default (106 registers, 4 waves)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04v; mkdir -p $O
for mw in 1 5 6 8; do
  RMEM_HIPCC_FLAGS="-DDW_MINW=$mw" python -m rmem_amd.build --force > $O/build_$mw.log 2>&1
  echo -n "minw $mw: "; timeout 300 python tools/kbench_gemm.py 2>> $O/kg.err | grep -E "dwconv" | tr '\n' ' '; echo
done | tee $O/r04v_dwconv_occupancy.txt
python -m rmem_amd.build --force > /dev/null 2>&1
