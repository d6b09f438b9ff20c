mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r02_p}
timeout 900 python -m pytest tests/test_hip_batched.py -x -q -s > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest.log; tail -4 gpurun_out/${TAG}_pytest.log
run() { # B KS
  RMEM_KS=$2 timeout 600 python bench.py --batched --clips-per-gpu $1 --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('B',c['clips_per_gpu'],'ks',c['key_splits_long_win_self'],'fps %.1f'%d['value'],'ms/step %.2f'%d['ms_per_step'])"
}
run 4 ""
run 4 "7,2,6"
run 4 "5,2,6"
run 4 "4,2,4"
run 4 "6,1,6"
run 4 "3,1,6"
run 2 ""
run 2 "10,3,6"
run 8 ""
run 8 "3,1,3"
run 8 "2,1,2"
