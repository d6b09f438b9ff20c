"""Which host-side operation keeps ROCr's AsyncEventsLoop thread spinning?  Each variant keeps the GPU busy for ~2 s;
prints wall, CPU seconds of the main thread and of the hottest other thread."""
import ctypes, os, sys, time, threading
import torch
hip = ctypes.CDLL("libamdhip64.so"); hip.hipSetDeviceFlags(4)
torch.cuda.init()
dev = "cuda"
x = torch.zeros(64 << 20, device=dev)          # 256 MB: x.add_ takes ~100 us
y = torch.zeros(64 << 20, device=dev)
pin = torch.zeros(1024, dtype=torch.uint8).pin_memory(); small = torch.zeros(1024, dtype=torch.uint8, device=dev)
s2 = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
x.add_(1); torch.cuda.synchronize()
with torch.cuda.graph(g):
    for _ in range(20):
        x.add_(1)
main_tid = str(threading.get_native_id())
def thr():
    tick = os.sysconf("SC_CLK_TCK"); out = {}
    for tid in os.listdir("/proc/self/task"):
        st = open(f"/proc/self/task/{tid}/stat").read()
        f = st[st.rindex(")") + 2:].split()
        out[tid] = (int(f[11]) + int(f[12])) / tick
    return out
def run(name, body, n):
    torch.cuda.synchronize()
    c0 = thr(); t0 = time.perf_counter()
    for i in range(n):
        body(i)
        if i % 50 == 49:                         # throttle: stay at most ~50 iterations ahead
            e = torch.cuda.Event(); e.record()
            if i >= 99: evs.pop(0).synchronize()
            evs.append(e)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0; c1 = thr()
    d = {k: v - c0.get(k, 0) for k, v in c1.items()}
    other = max(((v, k) for k, v in d.items() if k != main_tid), default=(0, ""))
    print(f"{name:34s} wall {wall:5.2f}  main {d.get(main_tid, 0):5.2f}  hottest other {other[0]:5.2f}")
evs = []
def kernels(i): x.add_(1)
def with_event(i):
    x.add_(1); e = torch.cuda.Event(); e.record()
def cross_stream(i):
    x.add_(1); e = torch.cuda.Event(); e.record()
    with torch.cuda.stream(s2):
        s2.wait_event(e); y.add_(1)
def cross_stream_back(i):
    x.add_(1); e = torch.cuda.Event(); e.record()
    with torch.cuda.stream(s2):
        s2.wait_event(e); y.add_(1); e2 = torch.cuda.Event(); e2.record()
    torch.cuda.current_stream().wait_event(e2)
def graph(i): g.replay()
def h2d(i):
    x.add_(1); small.copy_(pin, non_blocking=True)
def timing_event(i):
    x.add_(1); e = torch.cuda.Event(enable_timing=True); e.record()
for name, body, n in [("kernels only", kernels, 20000), ("+ event record", with_event, 20000), ("+ timing event record", timing_event, 20000),
                      ("+ cross-stream wait", cross_stream, 10000), ("+ cross-stream wait both ways", cross_stream_back, 10000),
                      ("graph replay (20 kernels)", graph, 1000), ("+ pinned H2D copy", h2d, 20000)]:
    evs.clear()
    run(name, body, n)
