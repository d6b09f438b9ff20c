"""Frames/s of multi-scale x flip test-time augmentation (scales 1.0 and 1.3, four augmentations) on a 480p clip:
one batched engine per image size (the default) against one engine per augmentation (RMEM_TTA=serial).
    python tools/tta_ms_probe.py [frames]"""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rmem_amd import driver as D                      # noqa: E402
from rmem_amd.config import get_config                # noqa: E402
from rmem_amd.model import build_vos_model            # noqa: E402
from rmem_amd.synth import load_synthetic_weights, synth_clip   # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 24
H, W = 481, 849
H2, W2 = D.restrict_size(480, 854, max_size=int(800 * 1.3), scale=1.3)
cfg = get_config("r50_deaotl", 1, 3)
model = build_vos_model(cfg.MODEL_VOS, cfg).eval()
load_synthetic_weights(model)
model = model.to("cuda:0")
imgs, lab = synth_clip(5, frames, H, W, 3)
imgs = [im.to("cuda:0") for im in imgs]
big = [F.interpolate(im, size=(H2, W2), mode="bicubic", align_corners=False) for im in imgs]
lab = lab.to("cuda:0")


def clip():
    return [D.make_samples(imgs[t], lab if t == 0 else None, (480, 854), 3, flip_aug=True, name=f"{t:05d}.jpg",
                           scaled_imgs=[big[t]]) for t in range(frames)]


out = {}
for tta in ("batched", "serial"):
    os.environ["RMEM_TTA"] = tta
    drv = D.ClipDriver(model, cfg)
    drv.run_clip(clip(), num_frames=frames)          # warm: graphs, solver choices
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = drv.run_clip(clip(), num_frames=frames)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out[tta] = res
    print(f"{tta}: {frames - 1} frames x 4 augmentations ({H}x{W}, {H2}x{W2}) in {dt * 1e3:.1f} ms = {(frames - 1) / dt:.1f} frames/s"
          f" ({4 * (frames - 1) / dt:.1f} engine-frames/s)", flush=True)
mism = [int((out["batched"].masks[i] != out["serial"].masks[i]).sum()) for i in range(frames - 1)]
print("batched vs serial mismatching pixels per frame:", mism)
