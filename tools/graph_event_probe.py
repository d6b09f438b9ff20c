#!/usr/bin/env python
"""Do external HIP events recorded INSIDE a captured hipGraph time a kernel of a replay?
(torch.cuda.Event(enable_timing=True, external=True) -> hipEventRecordWithFlags(hipEventRecordExternal) ->
event-record nodes.)  Prints the elapsed time per replay next to the eager measurement of the same kernel and
the cost of the two extra nodes per replay."""
import json
import sys
import time

import torch

dev = torch.device("cuda", 0)
a = torch.randn(4096, 4096, device=dev, dtype=torch.float16)
b = torch.randn(4096, 4096, device=dev, dtype=torch.float16)
c = torch.empty_like(a)
small = torch.zeros(1024, device=dev)
for _ in range(3):
    torch.mm(a, b, out=c)
torch.cuda.synchronize()
res = {}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.mm(a, b, out=c); e1.record(); e1.synchronize()
res["eager_us"] = 1e3 * e0.elapsed_time(e1)


def build(with_events):
    g = torch.cuda.CUDAGraph()
    x0 = torch.cuda.Event(enable_timing=True, external=True)
    x1 = torch.cuda.Event(enable_timing=True, external=True)
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(10):
                small.add_(1.0)
            if with_events:
                x0.record(s)
            torch.mm(a, b, out=c)
            if with_events:
                x1.record(s)
            for _ in range(10):
                small.add_(1.0)
    return g, x0, x1


try:
    g, x0, x1 = build(True)
    per = []
    for _ in range(5):
        g.replay()
        torch.cuda.synchronize()
        per.append(1e3 * x0.elapsed_time(x1))
    res["in_graph_us"] = per
    g2, _, _ = build(False)
    for name, gg in (("with_events", g), ("without", g2)):
        for _ in range(5):
            gg.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            gg.replay()
        torch.cuda.synchronize()
        res[f"replay_us_{name}"] = 1e6 * (time.perf_counter() - t0) / 200
except Exception as e:  # noqa: BLE001
    res["error"] = repr(e)
print(json.dumps(res))
