mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "read_" 2>&1 | tail -2
for ks in 7,2,9 6,3,9 6,3,6 5,4,6 7,2,6 6,3,4; do
  echo "KS=$ks"; RMEM_KS=$ks timeout 300 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['roofline']['isolated_mean_us'],1), round(d['roofline']['mean_us'],1))"
done
