cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, call ad: probe -- MIOpen's fused conv + bias (+ residual) + ReLU in the ResNet-50 encoder instead of conv -> HIP pass
O=$PWD/gpurun_out/r06ad; mkdir -p $O
run() { RMEM_ENC_FUSED=$1 RMEM_BENCH_KERNELS=0 timeout 400 python bench.py --no-dropin --cpu-frames 2 2>>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d.get('mask_mismatch_px'), (d.get('parity_vs_reference') or {}).get('mask_mismatch_px_total'))"; }
for rep in 1 2; do echo "conv + HIP pass $(run 0)   MIOpen fused $(run 1)"; done 2>&1 | tee $O/ab_enc_fused.txt
tail -3 $O/err.txt
