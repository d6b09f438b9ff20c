#!/usr/bin/env python
"""Does an engine give the same bytes when the same clip runs twice through it (same process)?
Prints, per frame, the first stage whose sha differs between run 1 and run 2: encoder features, LSTT
output, decoder logits; for the batched engine (B = 2) and the single-clip engine."""
import hashlib
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rmem_amd.batched import BatchedDeAOTEngine
from rmem_amd.config import get_config
from rmem_amd.engine import build_engine
from rmem_amd.model import build_vos_model
from rmem_amd.synth import load_synthetic_weights, synth_clip

DEV = "cuda:0"
hh = lambda t: hashlib.sha256(t.detach().float().cpu().numpy().tobytes()).hexdigest()[:10]
cfg = get_config("r50_deaotl", 1, 3)
model = build_vos_model("deaot", cfg).eval()
load_synthetic_weights(model)
model = model.to(DEV)
B, frames, Hh, Ww = 2, 6, 97, 129
clips = [synth_clip(600 + i, frames, Hh, Ww, 3) for i in range(B)]
TEACH = {}
LOG = []

from rmem_amd.batched import BatchedLSTT
from rmem_amd.lstt import DeAOTLSTT


def _wrap(cls, batched):
    orig = cls.forward

    def fwd(self, emb, ref_frame=False):
        c0 = self.clips[0] if batched else self
        pre = [("emb_in", hh(emb)), ("idemb", hh(c0.idemb_pl.hi)), ("ref", str(ref_frame))]
        if not ref_frame and c0.short is not None:
            pre += [("shortK_l0", hh(c0.bankK[0].hi[c0.short])), ("shortV_l0", hh(c0.bankV[0].hi[c0.short])),
                    ("shortV_l2", hh(c0.bankV[2].hi[c0.short]))]
        out = orig(self, emb, ref_frame)
        c0 = self.clips[0] if batched else self
        pre += [("Qpe", hh(c0.Qpe.hi)), ("G_main", hh(c0.ws_main.G)), ("out", hh(out))]
        LOG.append(pre)
        return out
    cls.forward = fwd


_wrap(BatchedLSTT, True)
_wrap(DeAOTLSTT, False)


def run(eng, kind, teach_key):
    rec = []
    eng.restart_engine()
    if kind == "batched":
        eng.add_reference_frame(torch.cat([c[0][0] for c in clips]).to(DEV), torch.cat([c[1] for c in clips]).to(DEV),
                                obj_nums=[3] * B, frame_step=0)
    else:
        eng.add_reference_frame(clips[0][0][0].to(DEV), clips[0][1].to(DEV), obj_nums=[3], frame_step=0)
    labs = []
    for t in range(1, frames):
        img = torch.cat([c[0][t] for c in clips]).to(DEV) if kind == "batched" else clips[0][0][t].to(DEV)
        lg = eng.match_propogate_one_frame(img, output_size=(Hh, Ww))
        lstt = eng.lstt if kind == "batched" else eng.aot_engines[0].lstt
        pl = eng.pred_id_logits if kind == "batched" else eng.aot_engines[0].pred_id_logits
        rec.append((hh(lstt.tgt if kind == "batched" else lstt.tgt), hh(lstt.out), hh(pl), hh(lg)))
        lab = lg.argmax(1, keepdim=True).float()
        labs.append(lab)
        fed = TEACH[teach_key][t - 1] if teach_key in TEACH else lab       # teacher-forced from the first run: differences do not feed back
        eng.update_memory(F.interpolate(fed, size=eng.input_size_2d, mode="nearest"))
    TEACH.setdefault(teach_key, labs)
    return rec


model.optimize_for_inference(True)
with torch.no_grad():
    for name, x in (("batch 1", clips[0][0][1].to(DEV)), ("batch 2", torch.cat([clips[0][0][1], clips[1][0][1]]).to(DEV))):
        outs = [[hh(f) for f in model.encode_image(x)] for _ in range(5)]
        print("encoder", name, "call k vs call 1 equal:", [o == outs[0] for o in outs], "vs previous call:",
              [outs[k] == outs[k - 1] for k in range(1, 5)], "levels of call 2 vs 1:", [a == b for a, b in zip(outs[0], outs[1])])

for kind in ("batched", "single"):
    if kind == "batched":
        eng = BatchedDeAOTEngine(model, B, long_term_mem_gap=2)
    else:
        eng = build_engine("deaotengine", phase="eval", aot_model=model, gpu_id=0, long_term_mem_gap=2)
        eng.eval()
    runs, logs = [], []
    for _ in range(3):
        del LOG[:]
        runs.append(run(eng, kind, kind))
        logs.append([list(x) for x in LOG])
    for f, (a, b) in enumerate(zip(logs[1], logs[2])):
        d = [n for (n, x), (_, y) in zip(a, b) if x != y]
        print(kind, f"run3 vs run2 LSTT call {f}:", "identical" if not d else "DIFFERS at " + ", ".join(d))
    for r in (1, 2):
        for f, (a, b) in enumerate(zip(logs[0], logs[r])):
            d = [n for (n, x), (_, y) in zip(a, b) if x != y]
            print(kind, f"run{r + 1} vs run1 LSTT call {f}:", "identical" if not d else "DIFFERS at " + ", ".join(d))
    for r in (1, 2):
        for t, (a, b) in enumerate(zip(runs[0], runs[r])):
            names = ["lstt_residual_out", "lstt_out", "decoder_logits", "upsampled"]
            diff = [n for n, x, y in zip(names, a, b) if x != y]
            print(kind, f"run{r + 1} vs run1 frame {t + 1}:", "identical" if not diff else "DIFFERS at " + ", ".join(diff))
