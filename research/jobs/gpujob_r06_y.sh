cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, call y: kernel trace of the AOT-block benches (R50-AOTL, SwinB-AOTL): where their frames go
export TMPDIR=/tmp
O=gpurun_out/r06y; mkdir -p $O
for m in r50_aotl; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -o $m -- python bench.py --model $m --steps 60 --warmup 5 --no-cpu-baseline --no-dropin > $O/prof_$m.log 2>&1
  python tools/prof_summary.py $O/prof_$m/${m}_kernel_trace.csv 30 > $O/r06y_bench_${m}_kernel_stats.md
  head -45 $O/r06y_bench_${m}_kernel_stats.md
  find $O/prof_$m -name "*.csv" -size +1M -delete
done
