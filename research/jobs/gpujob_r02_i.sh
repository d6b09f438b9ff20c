for ks in 6,3,6 6,3,9 7,2,9 5,4,9 6,2,9 5,3,9 6,3,12; do
  echo "KS=$ks"; RMEM_KS=$ks timeout 300 python bench.py --no-cpu-baseline --no-dropin --steps 60 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['roofline']['isolated_mean_us'],1), round(d['roofline']['mean_us'],1))"
done
