mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_engine.py -m gpu -x -q -k "read_ or lstt_forward or small_clip or paired" 2>&1 | tail -3
timeout 300 python tools/kbench.py > gpurun_out/r02_h_kbench.json 2>/dev/null; head -12 gpurun_out/r02_h_kbench.json
for ks in 7,2,6 6,3,6; do
  echo "KS=$ks"; RMEM_KS=$ks timeout 300 python bench.py --no-cpu-baseline --no-dropin --steps 60 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['roofline']['isolated_mean_us'],1), round(d['roofline']['mean_us'],1))"
done
