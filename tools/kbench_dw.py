#!/usr/bin/env python
"""Isolated timing of the depth-wise 5x5 -> planes kernel (one map / two maps per launch) at the
480p geometry; RMEM_DW="rx,v" selects the variant (one process per variant)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rmem_amd import hip

def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best

def main():
    h, w, C = int(os.environ.get("H", 31)), int(os.environ.get("W", 54)), 1024
    N = h * w; Np = (N + 127) // 128 * 128
    dev = "cuda:0"; lib = hip.load(); st = hip.stream_ptr()
    g0, g1 = torch.randn(N, C, device=dev), torch.randn(N, C, device=dev)
    w0, w1 = torch.randn(25, C, device=dev), torch.randn(25, C, device=dev)
    o = [torch.zeros(Np, C, dtype=torch.int16, device=dev) for _ in range(4)]
    one = lambda: hip.check(lib.rmem_dwconv5x5_split(g0.data_ptr(), C, w0.data_ptr(), h, w, C, o[0].data_ptr(), o[1].data_ptr(), C, st), "dw")
    two = lambda: hip.check(lib.rmem_dwconv5x5_split2(g0.data_ptr(), g1.data_ptr(), C, w0.data_ptr(), w1.data_ptr(), h, w, C,
                                                      o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), C, st), "dw2")
    one(); torch.cuda.synchronize()
    chk = int(o[0].long().sum().item()) ^ int(o[1].long().sum().item())
    print(json.dumps({"RMEM_DW": os.environ.get("RMEM_DW", "default"), "RMEM_DW_ROWS": os.environ.get("RMEM_DW_ROWS", "0"),
                      "one_us": round(timeit(one), 2), "two_us": round(timeit(two), 2), "checksum": chk}))
    if "--rows" in sys.argv:                   # the RY-rows-per-thread variants in the same process (the library reads the switch per launch)
        for ry in ("0", "2", "3", "4"):
            hip.configure("dw_rows", int(ry))
            one(); torch.cuda.synchronize()
            c2 = int(o[0].long().sum().item()) ^ int(o[1].long().sum().item())
            print(json.dumps({"RMEM_DW_ROWS": ry, "one_us": round(timeit(one), 2), "two_us": round(timeit(two), 2), "checksum": c2}))

main()
