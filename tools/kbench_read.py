#!/usr/bin/env python
"""Micro-benchmark of the fused memory read (rmem_attn_read + rmem_attn_read_combine) at the
480p K=4 / 720p K=8 problem sizes, isolated, HIP-event timed: long-term bank read, self read
(T=1) and windowed short-term read, for a sweep of key splits.

    python tools/kbench_read.py [--h 31 --w 54 --T 4] [--splits 5,7,9,12]
"""
import argparse
import ctypes as C
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=31)
    ap.add_argument("--w", type=int, default=54)
    ap.add_argument("--T", type=int, default=4)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--splits", default="4,9")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    from rmem_amd import hip
    lib = hip.load()
    dev = torch.device("cuda:0")
    h, w, T = args.h, args.w, args.T
    N = h * w
    Np = (N + 127) // 128 * 128
    S = T + 2
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g)
    K = hip.Planes.from_f32((rnd(S, Np, 128) * 1.5).to(dev))
    V = hip.Planes.from_f32(rnd(S, Np // 16, 1024, 16).to(dev))
    Q = hip.Planes.from_f32((rnd(Np, 128) * 1.5).to(dev))
    bias = (rnd(N, T) * 3).to(dev)
    U = rnd(N, 1024).to(dev)
    R = torch.zeros(N, 232, device=dev)
    R[:, :225] = rnd(N, 225).to(dev)
    smap = torch.arange(16, dtype=torch.int32, device=dev)
    G = torch.zeros(N, 1024, device=dev)
    mass = torch.zeros(N, T, device=dev)
    st = hip.stream_ptr()
    res = {"N": N, "T": T}
    maxs = max(int(x) for x in args.splits.split(","))
    part = torch.zeros(maxs, Np, 1024, device=dev)
    ml = torch.zeros(maxs, Np, 2, device=dev)
    lslot = torch.zeros(maxs, Np, T, 2, device=dev)

    def mk(mode, Tn, ks, want_mass):
        ra = hip.ReadArgs()
        ra.mode, ra.qh, ra.ql = mode, Q.hi.data_ptr(), Q.lo.data_ptr()
        ra.kh, ra.kl, ra.k_slot_stride = K.hi.data_ptr(), K.lo.data_ptr(), Np * 128
        ra.vh, ra.vl, ra.v_slot_stride = V.hi.data_ptr(), V.lo.data_ptr(), 1024 * Np
        ra.slot_map = smap.data_ptr()
        ra.T, ra.N, ra.Npad, ra.ncols, ra.scale = Tn, N, Np, 1024, 1.0 / math.sqrt(128)
        ra.bias = bias.data_ptr() if mode == 0 and Tn == T else None
        ra.R, ra.ldr = (R.data_ptr() if mode == 1 else None), 232
        ra.h, ra.w, ra.ksplits = h, w, ks
        ra.part, ra.ml = part.data_ptr(), ml.data_ptr()
        ra.lslot = lslot.data_ptr() if want_mass else None
        ca = hip.ReadCombineArgs()
        ca.T, ca.N, ca.Npad, ca.ncols, ca.ksplits = Tn, N, Np, 1024, ks
        ca.part, ca.ml, ca.lslot = part.data_ptr(), ml.data_ptr(), (lslot.data_ptr() if want_mass else None)
        ca.U, ca.ldu, ca.G, ca.ldg = U.data_ptr(), 1024, G.data_ptr(), 1024
        ca.mass = mass.data_ptr() if want_mass else None
        return ra, ca

    for name, mode, Tn, wm in (("long", 0, T, True), ("self", 0, 1, False), ("window", 1, 1, False)):
        if args.only and name not in args.only:
            continue
        for ks in [int(x) for x in args.splits.split(",")]:
            ra, ca = mk(mode, Tn, ks, wm)
            t_read = timeit(lambda: hip.check(lib.rmem_attn_read(C.byref(ra), st), "read"), args.iters)
            t_comb = timeit(lambda: hip.check(lib.rmem_attn_read_combine(C.byref(ca), st), "comb"), args.iters)

            def both():
                hip.check(lib.rmem_attn_read(C.byref(ra), st), "read")
                hip.check(lib.rmem_attn_read_combine(C.byref(ca), st), "comb")
            t_both = timeit(both, args.iters)
            res[f"{name}_ks{ks}"] = {"read_us": round(t_read, 2), "combine_us": round(t_comb, 2), "both_us": round(t_both, 2)}
        if name == "long":
            flops = 2.0 * N * (T * N) * (1024 + 128)
            best = min(v["read_us"] for k, v in res.items() if k.startswith("long_ks"))
            res["long_best_read_us"] = best
            res["long_algorithmic_TFLOPs"] = round(flops / (best * 1e-6) / 1e12, 1)
    # phase stamps of the long read (debug aid of the kernel: bank mode with R != NULL)
    ks = 9
    ra, ca = mk(0, T, ks, True)
    nblk = 8 * ((((Np // 128) * 2 * ks) + 7) // 8)
    tr = torch.zeros(nblk, 16, dtype=torch.int64, device=dev)
    ra.R = tr.data_ptr()
    for _ in range(3):
        hip.check(lib.rmem_attn_read(C.byref(ra), st), "read")
    torch.cuda.synchronize()
    t = tr.cpu().double()
    live = t[:, 3] > 0
    t = t[live]
    res["trace_ks9_cycles"] = {"blocks": int(live.sum()), "prologue": float((t[:, 1] - t[:, 0]).mean()),
                               "loop": float((t[:, 2] - t[:, 1]).mean()), "flush": float((t[:, 3] - t[:, 2]).mean()),
                               "per_tile": {n: float(t[:, 4 + k].mean()) / 12 for k, n in
                                            enumerate(["score", "softmax", "barrierA", "pv", "barrierB"])}}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
