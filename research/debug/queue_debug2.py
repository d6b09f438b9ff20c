import copy, sys, torch
sys.path.insert(0, ".")
from rmem_amd import driver as D
from rmem_amd.synth import synth_clip, load_synthetic_weights
from rmem_amd.config import get_config
from rmem_amd.model import build_vos_model
DEV = "cuda:0"
cfg = get_config("r50_deaotl", 1, 3)
model = build_vos_model("deaot", cfg).eval(); load_synthetic_weights(model); model = model.to(DEV)
Hh, Ww = 97, 129
def clip(cid, n):
    imgs, lab = synth_clip(500 + cid, n, Hh, Ww, 3)
    return [D.make_samples(imgs[t].to(DEV), lab.to(DEV) if t == 0 else None, (Hh, Ww), 3, name=f"{t:05d}.jpg") for t in range(n)]
a, b, d = clip(0, 7), clip(1, 5), clip(3, 4)
def hook(drv, log):
    eng = drv.engine
    orig = eng.match_propogate_one_frame
    def f(*args, **kw):
        enc_in = args[0].clone()
        up = orig(*args, **kw)
        l = eng.lstt
        c = l.clips[0]
        log.append(dict(img=enc_in, out=l.out.clone(), tgt=l.tgt.clone(), lab=l.label_buffer(*eng.input_size_2d).clone(),
                        idemb=c.idemb_pl.hi.clone(), k0=c.bankK[0].hi[c.cur].clone(), v0=c.bankV[0].hi[c.cur].clone(), cur=c.cur, T=c._T,
                        maps=c.maps.clone(), up=up.clone(), par=eng._par))
        return up
    eng.match_propogate_one_frame = f
F_, U_ = [], []
drvF = D.BatchedClipDriver(model, 2, cfg); hook(drvF, F_)
rF = drvF.run_queue([a, b, d])
drvU = D.BatchedClipDriver(model, 2, cfg)
rag = drvU.run_clips([a, b]); hook(drvU, U_)
rU = drvU.run_queue([a, b, d])
print("rag vs F", [int((rF[0].masks[t] != rag[0].masks[t]).sum()) for t in range(6)]); print("masks F vs U", [int((rF[0].masks[t] != rU[0].masks[t]).sum()) for t in range(6)])
for k in range(len(F_)):
    print("step", k, {n: (float((F_[k][n].float() - U_[k][n].float()).abs().max()) if torch.is_tensor(F_[k][n]) else (F_[k][n], U_[k][n])) for n in F_[k]})
