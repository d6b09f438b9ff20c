#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03q; mkdir -p $O; rm -f $O/mha_persist.txt
for p in 0 1 2 3 4 6; do
RMEM_MHA_PERSIST=$p timeout 300 python tools/kbench_mha.py 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('persist=$p', {k:v['us'] for k,v in d.items() if isinstance(v,dict)})" >> $O/mha_persist.txt
done
RMEM_MHA_PERSIST=3 timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_aot.py -q -m gpu -k "mha or aot" 2>&1 | tail -1 >> $O/mha_persist.txt
