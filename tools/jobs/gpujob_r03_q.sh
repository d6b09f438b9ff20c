#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03q; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_aot.py -q -m gpu -k "mha or aot or swin" 2>&1 | tail -2
timeout 300 python tools/kbench_mha.py > $O/kbench_mha.json 2>/dev/null; python -c "
import json
d=json.load(open('$O/kbench_mha.json')); print({k:v['us'] for k,v in d.items() if isinstance(v,dict)})"
