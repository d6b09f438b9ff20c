#!/usr/bin/env python
"""Micro-benchmark of the fused memory read (rmem_attn_read + rmem_attn_read_combine) at the
480p K=4 / 720p K=8 problem sizes, isolated, HIP-event timed: long-term bank read, self read
(T=1) and windowed short-term read, for a sweep of key splits; phase stamps of the kernel
(rmem_attn_read_trace); a numerical cross-check of every configuration against fp64 is NOT done here
(tests/test_hip_ops.py), only finiteness.

    python tools/kbench_read.py [--h 31 --w 54 --T 4] [--splits 4,6,9]
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
    ap.add_argument("--splits", default="6,9")
    ap.add_argument("--win-splits", default="1,2")
    ap.add_argument("--only", default="")
    ap.add_argument("--no-trace", action="store_true", help="timing only (for rocprofv3 --pmc runs)")
    ap.add_argument("--time-vars", default="", help="RMEM_READ_VAR values whose tracing-kernel launch is timed as well, e.g. 16")
    ap.add_argument("--old", action="store_true", help="also time the round-2 kernel if the library still has it (rmem_attn_read_v128)")
    args = ap.parse_args()
    from rmem_amd import hip
    lib = hip.load()
    old_read = getattr(lib, "rmem_attn_read_v128", None) if args.old else None
    if old_read is not None:
        old_read.restype = C.c_int
        old_read.argtypes = [C.POINTER(hip.ReadArgs), C.c_void_p]
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
    ls = [int(x) for x in args.splits.split(",")]
    ws = [int(x) for x in args.win_splits.split(",")]
    maxs = max(ls + ws)
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

    for name, mode, Tn, wm, sweep in (("long", 0, T, True, ls), ("self", 0, 1, False, ls), ("window", 1, 1, False, ws)):
        if args.only and name not in args.only:
            continue
        for ks in sweep:
            ra, ca = mk(mode, Tn, ks, wm)
            ent = {}
            ent["read_us"] = round(timeit(lambda: hip.check(lib.rmem_attn_read(C.byref(ra), st), "read"), args.iters), 2)
            ent["combine_us"] = round(timeit(lambda: hip.check(lib.rmem_attn_read_combine(C.byref(ca), st), "comb"), args.iters), 2)
            torch.cuda.synchronize()
            ent["finite"] = bool(torch.isfinite(G).all())
            if old_read is not None:
                ent["old_read_us"] = round(timeit(lambda: hip.check(old_read(C.byref(ra), st), "old"), args.iters), 2)
            if args.no_trace:
                res[f"{name}_ks{ks}"] = ent
                continue
            # phase stamps
            nblk = 8 * (((N + 63) // 64 * ks + 7) // 8)
            tr = torch.zeros(nblk, 64, dtype=torch.int64, device=dev)
            for var in ("4", "8", "0"):              # experiments of the tracing kernel (RMEM_READ_VAR); 0 = product, last
                hip.configure("read_var", int(var))
                tr.zero_()
                for _ in range(2):
                    hip.check(lib.rmem_attn_read_trace(C.byref(ra), tr.data_ptr(), st), "trace")
                torch.cuda.synchronize()
                tt = tr.cpu().double()
                tt = tt[tt[:, 3] > 0]
                ent[f"var{var}_unit_cycles_mean"] = round(float((tt[:, 3] - tt[:, 0]).mean()))
                ent[f"var{var}_loop_per_tile"] = round(float(((tt[:, 2] - tt[:, 1]) / tt[:, 28].clamp(min=1)).mean()))
                if var in ():
                    ntv = tt[:, 28:29].clamp(min=1)
                    ent[f"var{var}_top_score_pv_barrier_w0_w4"] = [[round(float((tt[:, o + w] / ntv[:, 0]).mean())) for o in (40, 4, 12, 20)] for w in (0, 4)]
            ent["trace_kernel_us"] = round(timeit(lambda: hip.check(lib.rmem_attn_read_trace(C.byref(ra), tr.data_ptr(), st), "trace"), args.iters), 2)
            for var in [v for v in args.time_vars.split(",") if v]:      # whole-launch time of an experiment variant (16: one MFMA per product)
                hip.configure("read_var", int(var))
                ent[f"trace_kernel_us_var{var}"] = round(timeit(lambda: hip.check(lib.rmem_attn_read_trace(C.byref(ra), tr.data_ptr(), st), "trace"), args.iters), 2)
            hip.configure("read_var", 0)
            t = tr.cpu().double()
            t = t[t[:, 3] > 0]
            if t.shape[0]:
                # launch shape from the device-wide 100 MHz counter: when units start and end relative to the first start
                st0 = t[:, 48].min()
                rel = lambda col: (t[:, col] - st0) / 100.0       # us
                ent["launch_shape_us"] = {
                    "last_start": round(float(rel(48).max()), 2), "median_start": round(float(rel(48).median()), 2),
                    "first_end": round(float(rel(49).min()), 2), "median_end": round(float(rel(49).median()), 2), "last_end": round(float(rel(49).max()), 2),
                    "unit_us_mean_max": [round(float(((t[:, 49] - t[:, 48]) / 100.0).mean()), 2), round(float(((t[:, 49] - t[:, 48]) / 100.0).max()), 2)],
                    "shader_cycles_per_us": round(float(((t[:, 3] - t[:, 0]) / ((t[:, 49] - t[:, 48]) / 100.0)).mean()), 1)}
                nt = t[:, 28:29].clamp(min=1)
                per = lambda a, b: [round(float(x)) for x in (t[:, a:b] / nt).mean(dim=0)]
                hw = tr.cpu()[tr.cpu()[:, 3] > 0][0, 32:40]
                ent["trace_cycles"] = {
                    "units": int(t.shape[0]), "tiles_per_unit": [float(nt.min()), float(nt.mean()), float(nt.max())],
                    "unit_total_mean_max": [float((t[:, 3] - t[:, 0]).mean()), float((t[:, 3] - t[:, 0]).max())],
                    "prologue": float((t[:, 1] - t[:, 0]).mean()),
                    "loop": float((t[:, 2] - t[:, 1]).mean()), "loop_per_tile": float(((t[:, 2] - t[:, 1]) / nt[:, 0]).mean()),
                    "stats_and_flush": float((t[:, 3] - t[:, 2]).mean()),
                    "score_per_tile_by_wave": per(4, 12), "pv_per_tile_by_wave": per(12, 20), "barrier_per_tile_by_wave": per(20, 28), "top_per_tile_by_wave": per(40, 48),
                    "simd_of_wave_block0": [int((int(x) >> 4) & 3) for x in hw], "cu_of_wave_block0": [int((int(x) >> 8) & 15) for x in hw]}
            res[f"{name}_ks{ks}"] = ent
        if name == "long":
            flops = 2.0 * N * (T * N) * (1024 + 128)
            best = min(v["read_us"] for k, v in res.items() if k.startswith("long_ks"))
            res["long_best_read_us"] = best
            res["long_algorithmic_TFLOPs"] = round(flops / (best * 1e-6) / 1e12, 1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
