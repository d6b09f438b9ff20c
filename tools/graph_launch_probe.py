"""Host cost of hipGraph launches (encoder graph / frame graph), alone and from two threads."""
import sys, time, threading; sys.path.insert(0, '.')
import torch
from rmem_amd.config import get_config
from rmem_amd.model import build_vos_model
from rmem_amd.synth import load_synthetic_weights, synth_clip
from rmem_amd.engine import build_engine
dev = torch.device('cuda:0')
cfg = get_config('r50_deaotl', 1, 3)
m = build_vos_model(cfg.MODEL_VOS, cfg).eval(); load_synthetic_weights(m); m = m.to(dev)
eng = build_engine(cfg.MODEL_ENGINE, phase='eval', aot_model=m, gpu_id=0, long_term_mem_gap=2)
imgs, lab = synth_clip(0, 8, 481, 849, 3); imgs = [i.to(dev) for i in imgs]; lab = lab.to(dev)
eng.add_reference_frame(imgs[0], lab, obj_nums=[3], frame_step=0)
import torch.nn.functional as F
for t in range(1, 14):
    lg = eng.match_propogate_one_frame(imgs[t % 8], output_size=(480, 854), next_img=imgs[(t + 1) % 8])
    pred = torch.argmax(lg, 1, keepdim=True).float()
    eng.update_memory(F.interpolate(pred, size=eng.input_size_2d, mode='nearest'))
torch.cuda.synchronize()
e = eng.aot_engines[0]
eg = list(e._eg.values())[0][0]
fg = list(e._fg.values())[-1][0]
ug = list(e._ug.values())[-1]
def host(g, n=20):
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); g.replay(); ts.append(time.perf_counter() - t0); torch.cuda.synchronize()
    return 1e3 * sorted(ts)[n // 2]
print("host ms per launch: encoder graph %.3f  frame graph %.3f  update graph %.3f" % (host(eg), host(fg), host(ug)))
# two threads
s2 = torch.cuda.Stream()
res = {}
def worker():
    torch.cuda.set_device(0)
    with torch.cuda.stream(s2):
        t0 = time.perf_counter()
        for _ in range(20): eg.replay()
        res['enc'] = (time.perf_counter() - t0) / 20 * 1e3
th = threading.Thread(target=worker); torch.cuda.synchronize()
t0 = time.perf_counter(); th.start()
for _ in range(20): fg.replay()
res['frame'] = (time.perf_counter() - t0) / 20 * 1e3
th.join(); torch.cuda.synchronize(); res['wall'] = (time.perf_counter() - t0) / 20 * 1e3
print("two threads, 20 launches each (ms per launch):", {k: round(v, 3) for k, v in res.items()})
t0 = time.perf_counter()
for _ in range(20): eg.replay(); fg.replay()
t1 = time.perf_counter(); torch.cuda.synchronize()
print("one thread, both per iteration: host %.3f ms, wall %.3f ms" % ((t1 - t0) / 20 * 1e3, (time.perf_counter() - t0) / 20 * 1e3))
