"""Frames/s of flip test-time augmentation on a 480p clip with one engine per augmentation (the AOT block always; the DeAOT
block under RMEM_TTA=serial): the engines on HIP streams of their own (default) against one stream (RMEM_TTA_STREAMS=0), and
the label maps of the two runs compared.
    python tools/tta_streams_probe.py [model] [frames]        # model: r50_aotl (default) | r50_deaotl | swinb_aotl"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rmem_amd import driver as D                      # noqa: E402
from rmem_amd.config import get_config                # noqa: E402
from rmem_amd.model import build_vos_model            # noqa: E402
from rmem_amd.synth import load_synthetic_weights, synth_clip   # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "r50_aotl"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 24
os.environ["RMEM_TTA"] = "serial"
cfg = get_config(name, 1, 3)
H, W = (481, 849) if cfg.MODEL_ALIGN_CORNERS else (480, 848)
model = build_vos_model(cfg.MODEL_VOS, cfg).eval()
load_synthetic_weights(model)
model = model.to("cuda:0")
imgs, lab = synth_clip(5, frames, H, W, 3)
imgs = [im.to("cuda:0") for im in imgs]
lab = lab.to("cuda:0")


def clip():
    return [D.make_samples(imgs[t], lab if t == 0 else None, (480, 854), 3, flip_aug=True, name=f"{t:05d}.jpg")
            for t in range(frames)]


out = {}
for streams in ("0", "1", "0", "1"):
    os.environ["RMEM_TTA_STREAMS"] = streams
    drv = D.ClipDriver(model, cfg, fixed_gap=5)
    drv.run_clip(clip(), num_frames=frames)          # warm: graphs, solver choices
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = drv.run_clip(clip(), num_frames=frames)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out[streams] = res
    print(f"{name} flip TTA, RMEM_TTA_STREAMS={streams}: {frames - 1} frames x 2 engines in {dt * 1e3:.1f} ms = {(frames - 1) / dt:.1f} frames/s",
          flush=True)
mism = [int((out["0"].masks[i] != out["1"].masks[i]).sum()) for i in range(frames - 1)]
print("one stream vs engine streams, mismatching pixels per frame:", mism)
