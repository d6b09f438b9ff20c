#!/usr/bin/env python
"""Isolated timing of rmem_id_assign on a realistic (piecewise-constant) and on a random label map;
RMEM_IDA = tokens per block (one process per variant).  Prints a checksum of the output planes."""
import os, sys, json, copy
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rmem_amd.config import get_config
from rmem_amd.model import build_vos_model
from rmem_amd.synth import load_synthetic_weights, synth_clip
from rmem_amd.lstt import DeAOTLSTT

def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best

cfg = get_config("r50_deaotl", 1, 3)
m = build_vos_model("deaot", cfg).eval(); load_synthetic_weights(m); m = m.to("cuda:0")
L = DeAOTLSTT(m, 31, 54, "cuda:0")
_, lab = synth_clip(0, 1, 481, 849, 3)
lab = lab[0, 0].to(torch.uint8).to("cuda:0").contiguous()
rnd = torch.randint(0, 11, (481, 849), dtype=torch.uint8, device="cuda:0")
out = {}
for name, l in (("synth_clip_label", lab), ("random_label", rnd)):
    L.assign_identity(l); torch.cuda.synchronize()
    out[name + "_checksum"] = int(L.idemb_pl.hi.long().sum().item()) * 31 + int(L.idemb_pl.lo.long().sum().item())
    out[name + "_us"] = round(timeit(lambda: L.assign_identity(l)), 2)
print(json.dumps({"RMEM_IDA": os.environ.get("RMEM_IDA", "1"), **out}))
