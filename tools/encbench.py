import sys, time, torch, copy
sys.path.insert(0, '.')
from rmem_amd.config import get_config
from rmem_amd.model import build_vos_model
from rmem_amd.synth import load_synthetic_weights
dev = 'cuda:0'
m = build_vos_model('deaot', get_config()).eval(); load_synthetic_weights(m); m = m.to(dev); m.optimize_for_inference()
x = torch.randn(1, 3, 481, 849, device=dev)
emb = torch.randn(1674, 512, device=dev)
def run(n, cl=False):
    xx = x.contiguous(memory_format=torch.channels_last) if cl else x
    with torch.no_grad():
        for _ in range(n):
            enc = m.encode_image(xx)
            lg = m.decode_id_logits(emb, enc)
    return lg
for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    for cl in (False, True):
        try:
            run(5, cl); torch.cuda.synchronize()
            t0 = time.perf_counter(); run(30, cl); torch.cuda.synchronize()
            print(f"benchmark={bench} channels_last={cl}: {(time.perf_counter()-t0)/30*1e3:.3f} ms / frame (enc+dec)")
        except Exception as e:
            print("fail", bench, cl, repr(e)[:200])
