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
def mm(x, y): return [int((x[t] != y[t]).sum()) for t in range(min(len(x), len(y)))]
drv0 = D.BatchedClipDriver(model, 2, cfg)
rag = drv0.run_clips([a, b])
for name, clips in [("queue[a,b]", [a, b]), ("queue[a,b,d]", [a, b, d]), ("queue[a,d]", [a, d]), ("queue[a,b] order given", [a, b])]:
    drv = D.BatchedClipDriver(model, 2, cfg)
    res = drv.run_queue(clips, order="given" if "given" in name else "longest_first")
    print(name, "fresh: a", mm(res[0].masks, rag[0].masks), "b" if clips[1] is b else "", mm(res[1].masks, rag[1].masks) if clips[1] is b else "", drv.queue_stats)
res = drv0.run_queue([a, b, d])
print("after run_clips on the same driver: a", mm(res[0].masks, rag[0].masks), "b", mm(res[1].masks, rag[1].masks))
res2 = drv0.run_queue([a, b, d])
print("repeat: a", mm(res2[0].masks, res[0].masks), "d", mm(res2[2].masks, res[2].masks))
dd = drv0.run_clips([d, d])
print("d vs lockstep", mm(res[2].masks, dd[0].masks))
