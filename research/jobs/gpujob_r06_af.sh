cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, call af: key splits at 720p K=8 chosen for the frame? (one round of units, fewer partial sets, windowed read in one split)
O=$PWD/gpurun_out/r06af; mkdir -p $O
run() { if [ "$1" = default ]; then unset RMEM_KS; else export RMEM_KS=$1; fi; RMEM_BENCH_KERNELS=0 timeout 600 python bench.py --config 720p_k8 --gap 2 --no-cpu-baseline --no-dropin 2>>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['value'],1), round(r.get('mean_us',0),1), r.get('kernel','')[:24])"; }
for rep in 1 2; do for ks in default 3,1,4 4,1,4 2,1,4 3,1,3 4,2,4; do echo "RMEM_KS=$ks $(run $ks)"; done; done 2>&1 | tee $O/ks_sweep_720p.txt
python - <<'PY'
import os
os.environ.pop("RMEM_KS", None)
import torch
from rmem_amd.config import get_config
from rmem_amd.lstt import DeAOTLSTT
from rmem_amd.model import build_vos_model
from rmem_amd.synth import load_synthetic_weights
cfg = get_config("r50_deaotl", 1, 7)
m = build_vos_model("deaot", cfg).eval(); load_synthetic_weights(m); m = m.to("cuda:0")
L = DeAOTLSTT(m, 46, 81, torch.device("cuda:0"))
print("default splits at 46x81, cap", cfg.mem_cap, ":", L.ks_long, L.ks_win, L.ks_self)
PY
