"""Run the same teacher-forced clip several times and report which stage first differs bitwise
between runs (encoder features, LSTT outputs, decoder logits).  GPU only."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from rmem_amd.config import get_config
from rmem_amd.model import build_vos_model
from rmem_amd.engine import build_engine
from rmem_amd.synth import load_synthetic_weights, synth_clip

DEV = "cuda:0"
H, W, FR = int(os.environ.get("PH", 97)), int(os.environ.get("PW", 129)), 12
cfg = get_config("r50_deaotl", 1, 3)
model = build_vos_model(cfg.MODEL_VOS, cfg)
load_synthetic_weights(model)
model = model.to(DEV).eval()
eng = build_engine(cfg.MODEL_ENGINE, phase="eval", aot_model=model, gpu_id=0, long_term_mem_gap=2)
imgs, lab = synth_clip(7, FR, H, W, 3)
imgs = [i.to(DEV) for i in imgs]
lab = lab.to(DEV)

def run(teach):
    eng.restart_engine()
    eng.add_reference_frame(imgs[0], lab, obj_nums=[3], frame_step=0)
    rec, preds = [], []
    for t in range(1, FR):
        logit = eng.match_propogate_one_frame(imgs[t], output_size=(H, W))
        e = eng.aot_engines[0]
        rec.append((e.lstt.out.clone(), e.pred_id_logits.clone(), logit.clone()))
        pred = torch.argmax(logit, dim=1, keepdim=True).float()
        preds.append(pred)
        fed = teach[t - 1] if teach else pred
        eng.update_memory(F.interpolate(fed, size=eng.input_size_2d, mode="nearest"))
    return rec, preds


with torch.no_grad():
    e0 = [f.clone() for f in model.encode_image(imgs[1])]
    for i in range(5):
        e1 = model.encode_image(imgs[1])
        d = max(float((a - b).abs().max()) for a, b in zip(e0, e1))
        if d:
            print("encoder run-to-run diff", d)
_, teach = run(None)
runs = [run(teach)[0] for i in range(5)]
base = runs[0]
for i, r in enumerate(runs[1:], 1):
    for t in range(FR - 1):
        ls = float((base[t][0] - r[t][0]).abs().max())
        lg = float((base[t][1] - r[t][1]).abs().max())
        up = float((base[t][2] - r[t][2]).abs().max())
        if ls or lg or up:
            print(f"run{i} frame{t+1}: lstt diff {ls:.3e} logit diff {lg:.3e} upsampled {up:.3e}")
import hashlib
hh = lambda t: hashlib.md5(t.detach().cpu().numpy().tobytes()).hexdigest()[:8]
print("enc hashes", [hh(f) for f in e0])
for t in range(FR - 1):
    print("frame", t + 1, "lstt", hh(base[t][0]), "logit", hh(base[t][1]))
print("done; RMEM_NO_GRAPHS =", os.environ.get("RMEM_NO_GRAPHS"))
