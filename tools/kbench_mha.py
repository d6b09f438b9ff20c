#!/usr/bin/env python
"""Isolated timing of the AOT flash MHA kernel (rmem_mha_flash, 8 heads x 32) at the 480p geometry:
long-term read (T slots) and self / short-term read (T = 1), for a sweep of key splits."""
import argparse, ctypes as C, json, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rmem_amd import hip


def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=1674)
    ap.add_argument("--T", type=int, default=4)
    ap.add_argument("--splits", default="4,6,8,12,16")
    ap.add_argument("--only-long", action="store_true")
    args = ap.parse_args()
    lib = hip.load(); dev = "cuda:0"; st = hip.stream_ptr()
    N, T = args.N, args.T
    Np = (N + 127) // 128 * 128
    g = torch.Generator().manual_seed(0)
    P = hip.Planes.from_f32
    q = P((torch.randn(Np, 256, generator=g) * 1.2).to(dev))
    k = P((torch.randn(T + 2, Np, 256, generator=g) * 1.2).to(dev))
    v = P(torch.randn(T + 2, 256, Np, generator=g).to(dev))
    bias = (torch.randn(N, 8, T, generator=g) * 2).to(dev)
    smap = torch.arange(16, dtype=torch.int32, device=dev)
    res = {"N": N, "T": T}
    for Tn in ((T,) if args.only_long else (T, 1)):
        for ks in [int(x) for x in args.splits.split(",")]:
            opart = torch.zeros(ks, Np, 256, device=dev); ml = torch.zeros(ks, Np, 8, 2, device=dev)
            sml = torch.zeros(ks, Np, 8, Tn, 2, device=dev)
            a = hip.MHAArgs()
            a.qh, a.ql, a.ldq = q.hi.data_ptr(), q.lo.data_ptr(), 256
            a.kh, a.kl, a.k_slot_stride, a.ldk = k.hi.data_ptr(), k.lo.data_ptr(), Np * 256, 256
            a.vh, a.vl, a.v_slot_stride, a.ldv = v.hi.data_ptr(), v.lo.data_ptr(), 256 * Np, Np
            a.slot_map, a.T, a.N, a.Npad, a.heads = smap.data_ptr(), Tn, N, Np, 8
            a.scale, a.bias, a.ksplits = 1 / math.sqrt(32), (bias.data_ptr() if Tn == T else None), ks
            a.opart, a.ml, a.slot_ml, a.nsplit = opart.data_ptr(), ml.data_ptr(), sml.data_ptr(), 3
            t = timeit(lambda: hip.check(lib.rmem_mha_flash(C.byref(a), st), "mha"))
            flops = 2.0 * N * Tn * N * 512
            res[f"T{Tn}_ks{ks}"] = {"us": round(t, 1), "TFLOPs": round(flops / t / 1e6, 1)}
    print(json.dumps(res))


main()
