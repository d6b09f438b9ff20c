# A/B with bench flags: bash research/jobs/gpujob_ab2.sh "<ENV=.. -- flags>" ...
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
for v in "$@"; do
  envs="${v%%--*}"; flags="${v#*--}"; [ "$flags" = "$v" ] && flags=""
  echo -n "[$v] "; env $envs python bench.py --no-cpu-baseline $flags 2>gpurun_out/ab2.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['mean_us'],1))" || tail -5 gpurun_out/ab2.err
done; done
