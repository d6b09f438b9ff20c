"""Probe: can the frame be captured into hipGraphs with torch.cuda.CUDAGraph?
(a) encoder + decoder (MIOpen / rocBLAS under capture), (b) LSTT forward incl. the side stream."""
import sys, time, torch, copy
sys.path.insert(0, '.')
from rmem_amd.config import get_config
from rmem_amd.model import build_vos_model
from rmem_amd.synth import load_synthetic_weights
from rmem_amd.lstt import DeAOTLSTT
dev = torch.device('cuda:0')
m = build_vos_model('deaot', get_config()).eval(); load_synthetic_weights(m); m = m.to(dev); m.optimize_for_inference()
x = torch.randn(1, 3, 481, 849, device=dev)
emb = torch.randn(1674, 512, device=dev)

def encdec():
    enc = m.encode_image(x)
    return enc, m.decode_id_logits(emb, enc)

with torch.no_grad():
    for _ in range(3):
        ref_enc, ref_lg = encdec()
    torch.cuda.synchronize()
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            encdec()
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            enc, lg = encdec()
        g.replay(); torch.cuda.synchronize()
        print("enc/dec capture OK; max diff", (lg - ref_lg).abs().max().item())
        t0 = time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize()
        print("enc+dec graph replay: %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
    except Exception as e:
        print("enc/dec capture FAILED:", repr(e)[:300])

# (b) LSTT forward device part
L = DeAOTLSTT(m, 31, 54, dev, nsplit=3)
lab = torch.randint(0, 4, (481, 849), dtype=torch.uint8, device=dev)
tok = torch.randn(1674, 256, device=dev)
with torch.no_grad():
    L.assign_identity(lab); L.forward(tok, ref_frame=True)
    for i in range(6):
        L.forward(tok); L.assign_identity(lab); L.update_short_memories(i % 2 == 0)
        if len(L.bank) > L.cap: del L.bank[1]
    torch.cuda.synchronize()
    ref = L.forward(tok).clone(); torch.cuda.synchronize()
    try:
        g2 = torch.cuda.CUDAGraph()
        L._prepare(False)
        with torch.cuda.graph(g2):
            L._forward_device(False)
        L.tgt.copy_(tok); g2.replay(); torch.cuda.synchronize()
        print("LSTT capture OK; max diff vs eager", (L.out - ref).abs().max().item())
        t0 = time.perf_counter()
        for _ in range(50): g2.replay()
        torch.cuda.synchronize()
        print("LSTT graph replay: %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
        t0 = time.perf_counter()
        for _ in range(50): L._forward_device(False)
        torch.cuda.synchronize()
        print("LSTT eager: %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
    except Exception as e:
        import traceback; traceback.print_exc()
        print("LSTT capture FAILED:", repr(e)[:300])
