#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04tta; mkdir -p $O
timeout 900 python -m pytest tests/test_driver.py -x -q -m gpu -s -k "hip_vs_reference_golden" --timeout 400 2>&1 | grep -v amdgpu | tail -12 | cut -c1-330
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu | tail -4
import os, sys, time, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from rmem_amd import driver as D
from rmem_amd.config import get_config
from rmem_amd.model import build_vos_model
from rmem_amd.synth import load_synthetic_weights, synth_clip
dev = "cuda:0"
cfg = get_config("r50_deaotl", 1, 3)
model = build_vos_model(cfg.MODEL_VOS, cfg).eval(); load_synthetic_weights(model); model = model.to(dev)
H, W, HO, WO, n = 465, 833, 480, 854, 40
imgs, lab = synth_clip(11, n, H, W, 3)
lab0 = F.interpolate(lab, size=(HO, WO), mode="nearest").to(dev)
frames = [D.make_samples(imgs[t].to(dev), lab0 if t == 0 else None, (HO, WO), 3, flip_aug=True, name=f"{t:05d}.jpg") for t in range(n)]
out = {}
for mode in ("serial", "batched"):
    os.environ["RMEM_TTA"] = mode
    drv = D.ClipDriver(model, cfg)
    drv.run_clip(frames, num_frames=n)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = drv.run_clip(frames, num_frames=n)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out[mode] = (round((n - 1) / dt, 1), r.masks)
print("flip TTA at 480p, frames/s: serial", out["serial"][0], "batched", out["batched"][0], "x", round(out["batched"][0] / out["serial"][0], 2))
print("mismatching pixels per frame, first 8:", [int((out["serial"][1][t] != out["batched"][1][t]).sum()) for t in range(8)])
PY
