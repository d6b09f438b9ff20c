# generic A/B: bash research/jobs/gpujob_ab.sh "<ENV=.. ENV=..>" "<ENV=..>" ...   (each variant benched twice, interleaved)
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
for v in "$@"; do
  echo -n "[$v] "; env $v python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['mean_us'],1))"
done; done
