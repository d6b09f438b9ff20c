"""Does a CU-masked stream confine kernels?  fp32 matmul 4096^3 on the default stream and on masked streams."""
import sys, time, torch
sys.path.insert(0, ".")
from rmem_amd.streams import masked_stream
dev = torch.device("cuda:0")
a = torch.randn(4096, 4096, device=dev); b = torch.randn(4096, 4096, device=dev)
def t(st):
    with torch.cuda.stream(st):
        for _ in range(3): a @ b
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(10): a @ b
        e1.record(st); e1.synchronize()
    return e0.elapsed_time(e1) / 10
print("default stream: %.3f ms" % t(torch.cuda.current_stream()))
for n in (192, 128, 64):
    st = masked_stream(dev, n)
    print("masked %3d CUs: %.3f ms" % (n, t(st)))
# graph captured on the default stream, replayed on a masked stream
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    c = a @ b
st = masked_stream(dev, 64)
for name, s in (("default", torch.cuda.current_stream()), ("masked 64", st)):
    with torch.cuda.stream(s):
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(10): g.replay()
        e1.record(s); e1.synchronize()
    print("graph replay on %s: %.3f ms" % (name, e0.elapsed_time(e1) / 10))
