"""Does a clip's result depend on the SLOT it sits in (480p, B = 8)?  Same clip in all 8 slots; then per stage:
encoder features, LSTT output, decoder logits of slot s against slot 0."""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from rmem_amd.determinism import reproducible_convolutions
reproducible_convolutions()
from rmem_amd import driver as D
from rmem_amd.synth import synth_clip, load_synthetic_weights
from rmem_amd.config import get_config
from rmem_amd.model import build_vos_model
DEV = "cuda:0"
cfg = get_config("r50_deaotl", 1, 3)
model = build_vos_model("deaot", cfg).eval(); load_synthetic_weights(model); model = model.to(DEV)
H_IN, W_IN, H_OUT, W_OUT = 465, 833, 480, 854
def clip(cid, n):
    imgs, lab = synth_clip(cid, n, H_IN, W_IN, 3)
    lab0 = F.interpolate(lab, size=(H_OUT, W_OUT), mode="nearest").to(DEV)
    return [D.make_samples(imgs[t].to(DEV), lab0 if t == 0 else None, (H_OUT, W_OUT), 3, name=f"{t:05d}.jpg") for t in range(n)]
c = clip(2003, 8)
drv = D.BatchedClipDriver(model, 8, cfg)
eng = drv.engine
log = []
orig = eng.match_propogate_one_frame
def f(*a, **k):
    up = orig(*a, **k)
    enc = eng._eg[(tuple(a[0].shape), eng._par)][2]
    log.append(dict(enc=[e.clone() for e in enc], out=eng.lstt.out.clone(), logit=eng.pred_id_logits.clone()))
    return up
eng.match_propogate_one_frame = f
res = drv.run_clips([c] * 8)
print("masks: slot s vs slot 0, mismatching pixels per frame")
for s in range(1, 8):
    print(s, [int((res[s].masks[t] != res[0].masks[t]).sum()) for t in range(7)])
st = log[0]
for s in range(1, 8):
    print("frame 1, slot", s, "enc max abs diff", [float((e[s] - e[0]).abs().max()) for e in st["enc"]],
          "lstt", float((st["out"][s] - st["out"][0]).abs().max()), "logits", float((st["logit"][s] - st["logit"][0]).abs().max()))
