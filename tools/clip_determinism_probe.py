#!/usr/bin/env python
"""Does a clip's label-map hash depend on what the process ran before it?  Runs the given clip ids in
order through one ClipDriver (HIP engines) and prints sha256 per clip: compare invocations."""
import hashlib, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rmem_amd import driver as D
from rmem_amd.config import get_config
from rmem_amd.model import build_vos_model
from rmem_amd.synth import load_synthetic_weights, synth_clip

ids = [int(x) for x in sys.argv[1].split(",")]
frames, H, W = 8, 97, 129
dev = "cuda:0"
cfg = get_config("r50_deaotl", 1, 3)
model = build_vos_model(cfg.MODEL_VOS, cfg).eval()
load_synthetic_weights(model)
model = model.to(dev)
drv = D.ClipDriver(model, cfg, fixed_gap=2)
out = {}
for cid in ids:
    imgs, lab = synth_clip(100 + cid, frames, H, W, 3)
    fr = [D.make_samples(imgs[t].to(dev), lab.to(dev) if t == 0 else None, (H, W), 3, name=f"{t:05d}.jpg") for t in range(frames)]
    res = drv.run_clip(fr, num_frames=frames)
    out[cid] = [hashlib.sha256(res.masks[i].cpu().numpy().tobytes()).hexdigest()[:8] for i in range(res.masks.shape[0])]
print(json.dumps(out))
