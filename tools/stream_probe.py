"""Which freshly created streams overlap with the default stream?  (HW-queue placement probe)"""
import sys, time; sys.path.insert(0, '.')
import torch
cur = torch.cuda.current_stream()
SP = 2_000_000
a = torch.randn(4096, 4096, device='cuda')
def wall(other, work):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(other):
        work()
    work()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
sleep = lambda: torch.cuda._sleep(SP)
mm = lambda: [a @ a for _ in range(4)]
for w, name in ((sleep, "sleep"), (mm, "matmul")):
    w(); torch.cuda.synchronize()
    print(name, "serial", round(wall(cur, w), 3))
    res = []
    streams = [torch.cuda.Stream() for _ in range(12)]
    for s in streams:
        wall(s, w)
        res.append(round(wall(s, w), 3))
    print(name, res)
