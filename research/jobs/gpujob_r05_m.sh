cd "${GRAFT_REPO_ROOT:-/root/repo}"
# A/B: layer 0's long-term read launched with the hoisted front part (RMEM_EARLY_LONG_READ=1) against the paired launch
for v in 0 1 0 1; do echo -n "EARLY_LONG_READ=$v "; RMEM_FORCE_DIST=1 RMEM_EARLY_LONG_READ=$v RMEM_BENCH_EAGER_SAMPLE=1 RMEM_BENCH_KERNELS=0 timeout 300 python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config'].get('gathered_masks_sha256','')[:16])"; done
