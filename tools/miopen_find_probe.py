"""Probe: the encoder + decoder passes (PyTorch-ROCm / MIOpen, fp32) with MIOpen's immediate-mode solver choice (the
default, torch.backends.cudnn.benchmark = False) against the searched choice (benchmark = True), captured in a hipGraph
and replayed with the GPU to itself.  python tools/miopen_find_probe.py default|benchmark [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rmem_amd.config import get_config                # noqa: E402
from rmem_amd.model import build_vos_model            # noqa: E402
from rmem_amd.synth import load_synthetic_weights     # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "default"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
torch.backends.cudnn.benchmark = (mode == "benchmark")
dev = "cuda:0"
cfg = get_config("r50_deaotl", 1, 3)
model = build_vos_model(cfg.MODEL_VOS, cfg).eval()
load_synthetic_weights(model)
model = model.to(dev).optimize_for_inference()
g = torch.Generator().manual_seed(0)
img = torch.randn(B, 3, 481, 849, generator=g).to(dev)
emb = torch.randn(31 * 54, 512, generator=g).to(dev)


def graph_time(fn, n=30):
    t0 = time.perf_counter()
    with torch.no_grad():
        for _ in range(3):
            out = fn()
        torch.cuda.synchronize()
        t_warm = time.perf_counter() - t0
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            out = fn()
    gr.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        gr.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3, t_warm, out


with torch.no_grad():
    us_e, warm_e, xs = graph_time(lambda: model.encode_image(img))
    xs1 = type(xs)([x[:1].contiguous() for x in xs])
    if getattr(xs, "adapters", None) is not None:
        xs1.adapters = [a[:1].contiguous() for a in xs.adapters]
    us_d, warm_d, lg = graph_time(lambda: model.decode_id_logits(emb, xs1))
import hashlib
print(f"{mode}: encoder batch {B} {us_e:.1f} us ({us_e / B:.1f} per frame; first calls {warm_e:.1f} s)   decoder {us_d:.1f} us (first calls {warm_d:.1f} s)"
      f"   feature sha {hashlib.sha256(xs[-1].cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
