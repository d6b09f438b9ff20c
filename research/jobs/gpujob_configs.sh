# the other BASELINE.json configurations with the current engine
for m in r50_aotl swinb_aotl; do
  echo -n "$m 480p_k4: "; python bench.py --no-cpu-baseline --model $m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), d['config']['host_issue_ms_per_step'])"
done
echo -n "r50_deaotl 720p_k8: "; python bench.py --no-cpu-baseline --config 720p_k8 --gap 2 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['frac'], d['roofline']['mean_us'])"
echo -n "r50_deaotl 480p x1: "; python bench.py --no-cpu-baseline --nsplit 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['frac'])"
