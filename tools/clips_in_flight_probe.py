"""Frames/s of n 480p clips served by InFlightClipDriver with 1, 2 and 3 lanes (one ClipDriver + HIP stream per lane).
    python tools/clips_in_flight_probe.py [model] [clips] [frames]"""
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
nclips = int(sys.argv[2]) if len(sys.argv) > 2 else 6
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 24
cfg = get_config(name, 1, 3)
H, W = (481, 849) if cfg.MODEL_ALIGN_CORNERS else (480, 848)
model = build_vos_model(cfg.MODEL_VOS, cfg).eval()
load_synthetic_weights(model)
model = model.to("cuda:0")
clips = []
for c in range(nclips):
    imgs, lab = synth_clip(40 + c, frames, H, W, 3)
    clips.append([D.make_samples(imgs[t].to("cuda:0"), lab.to("cuda:0") if t == 0 else None, (480, 854), 3, name=f"{t:05d}.jpg")
                  for t in range(frames)])
ref = None
for lanes in (1, 2, 3, 1, 2):
    drv = D.InFlightClipDriver(model, lanes, cfg, fixed_gap=5)
    drv.run_clips(clips[:lanes])                        # warm: graphs, solver choices
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = drv.run_clips(clips)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if ref is None:
        ref = res
    same = all(torch.equal(a.masks, b.masks) for a, b in zip(res, ref))
    print(f"{name}: {nclips} clips x {frames - 1} frames, {lanes} lane(s): {dt * 1e3:.1f} ms = {nclips * (frames - 1) / dt:.1f} frames/s"
          f"; label maps equal to the first run: {same}", flush=True)
