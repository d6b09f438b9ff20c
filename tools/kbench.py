#!/usr/bin/env python
"""Per-kernel micro-benchmark of the hot-path kernels at the 480p K=4 (or 720p K=8)
problem sizes, each timed in isolation with HIP events (median of `--iters` launches).
Used to iterate on kernel performance; results land in profiles/ when worth keeping."""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeit(fn, iters):
    """`iters` back-to-back launches captured in ONE hipGraph and replayed: per-launch time with the queue kept full
    (includes the kernel boundary, excludes the Python / ctypes cost of a launch, which is 8-30 us -- more than the
    small kernels themselves, and what the eagerly timed numbers of rounds 1-3 mostly measured for them).
    RMEM_KBENCH_EAGER=1 times eager launches as before."""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    if os.environ.get("RMEM_KBENCH_EAGER") == "1":
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
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=31)
    ap.add_argument("--w", type=int, default=54)
    ap.add_argument("--cap", type=int, default=4)
    ap.add_argument("--nsplit", type=int, default=3)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--only", default="", help="pv: run only the fused memory-read launches (for --pmc runs)")
    args = ap.parse_args()
    from rmem_amd import hip
    from rmem_amd.config import get_config
    from rmem_amd.lstt import DeAOTLSTT, temporal_pe_rows
    from rmem_amd.model import build_vos_model
    from rmem_amd.synth import load_synthetic_weights
    dev = torch.device("cuda:0")
    cfg = get_config("r50_deaotl", 1, args.cap - 1)
    model = build_vos_model("deaot", cfg).eval()
    load_synthetic_weights(model)
    model = model.to(dev)
    L = DeAOTLSTT(model, args.h, args.w, dev, nsplit=args.nsplit)
    N, Np, T = L.N, L.Npad, args.cap
    g = torch.Generator(device="cpu").manual_seed(0)
    for l in range(L.L):
        for pl in (L.bankK[l], L.bankV[l]):
            x = torch.randn(pl.hi.shape, generator=g) * 1.0
            p = hip.Planes.from_f32(x.to(dev))
            pl.hi.copy_(p.hi)
            pl.lo.copy_(p.lo)
    L.bank, L.short, L.cur = list(range(T)), T - 1, T
    L.tgt.normal_()
    L.tgt_id.normal_()
    L.maps.copy_(torch.tensor(list(range(16)) + [T - 1] + [0] * 15, dtype=torch.int32))
    L.Ucat.normal_()
    q = hip.Planes.from_f32(torch.randn(Np, 128, device=dev))
    L.Qpe.hi.copy_(q.hi); L.Qpe.lo.copy_(q.lo)
    W = L.lw[1]
    res = {}
    lib = hip.load()
    rows = (C.c_int32 * 16)(*(temporal_pe_rows(T) + [0] * (16 - T)))
    map_bank, map_short = L.maps.data_ptr(), L.maps.data_ptr() + 64
    curK = L.bankK[1][T]
    # --- the three memory reads of a layer through the executor's own argument builders
    L._layer = 1
    sQK = hip.Planes(L.selfQK.hi[0], L.selfQK.lo[0])
    A = L._read_args(L.ws_main, 0, T, L.bankK[1], L.bankV[1], map_bank, L.Qpe, L.bias_pe, L.Ucat, True, L.ks_long)
    B = L._read_args(L.ws_side, 1, 1, L.bankK[1], L.bankV[1], map_short, curK, None, L.Ucat, False, L.ks_win)
    S = L._read_args(L.ws_main, 0, 1, L.selfQK, L.selfV, None, sQK, None, L.Uself, False, L.ks_self)
    if args.only == "pv":       # for --pmc runs: only the fused read launches
        for _ in range(args.iters):
            hip.check(lib.rmem_attn_read2(C.byref(A[0]), C.byref(B[0]), hip.stream_ptr()), "read2")
            hip.check(lib.rmem_attn_read(C.byref(S[0]), hip.stream_ptr()), "read")
        torch.cuda.synchronize()
        return
    res["ks_long"], res["ks_win"], res["ks_self"] = L.ks_long, L.ks_win, L.ks_self
    res["read2_long+window"] = timeit(lambda: hip.check(lib.rmem_attn_read2(C.byref(A[0]), C.byref(B[0]), hip.stream_ptr()), "r2"), args.iters)
    res["read_combine2"] = timeit(lambda: hip.check(lib.rmem_attn_read_combine2(C.byref(A[1]), C.byref(B[1]), hip.stream_ptr()), "c2"), args.iters)
    res["read_long_alone"] = timeit(lambda: hip.check(lib.rmem_attn_read(C.byref(A[0]), hip.stream_ptr()), "r"), args.iters)
    res["read_window_alone"] = timeit(lambda: hip.check(lib.rmem_attn_read(C.byref(B[0]), hip.stream_ptr()), "r"), args.iters)
    res["read_self"] = timeit(lambda: hip.check(lib.rmem_attn_read(C.byref(S[0]), hip.stream_ptr()), "r"), args.iters)
    res["read_combine_self"] = timeit(lambda: hip.check(lib.rmem_attn_read_combine(C.byref(S[1]), hip.stream_ptr()), "c"), args.iters)
    res["read2_TFLOPs_algorithmic"] = L.read_flops(T) / res["read2_long+window"] / 1e6
    if args.only == "reads":
        print(json.dumps({k: round(v, 2) for k, v in res.items()}, indent=1))
        return
    res["dwconv"] = timeit(lambda: L._dwconv(L.ws_main, W.dw_lt, L.Ylt), args.iters)
    ns = L.nsplit
    res["ln"] = timeit(lambda: L._ln(L.tgt, W.ln1, L.x_pl, 256), args.iters)
    res["gemm_Q(128x256)"] = timeit(lambda: hip.linear(L.x_pl, W.Wq, N, 128, 256, ldx=256, ldy=256, bias=W.bq,
                                                       pa=curK, ldpa=128, pb=L.Qpe,
                                                       ldpb=128, addvec=L.cur_pe, nsplit=ns), args.iters)
    curV = L.bankV[1][T]
    res["gemm_V(512x256,blocked-16 out)"] = timeit(lambda: hip.linear(
        L.x_pl, W.Wv, N, 512, 256, ldx=256, ldy=256, bias=W.bv, act=1, pa=curV, ldpa=1024, pa_blocked=True,
        nsplit=ns), args.iters)
    res["gemm_U(512x256)"] = timeit(lambda: hip.linear(L.x_pl, W.Wu, N, 512, 256, ldx=256, ldy=256, bias=W.bu, act=1,
                                                       d0=L.Ucat.data_ptr(), ldd0=1024, nsplit=ns), args.iters)
    res["gemm_proj_ls(512x2048)"] = timeit(lambda: hip.linear(
        L.Ylt, W.Wp_ls, N, 512, 2048, ldx=1024, ldy=2048, x2=L.Yst, ldx2=1024, kx_split=1024, bias=W.bp_ls,
        d0=L.tgt.data_ptr(), ldd0=256, d1=L.tgt_id.data_ptr(), ldd1=256, csplit=256, accumulate=True, nsplit=ns),
        args.iters)
    res["gemm_proj_self(512x1024)"] = timeit(lambda: hip.linear(
        L.Ylt, W.Wp_self, N, 512, 1024, ldx=1024, ldy=1024, bias=W.bp_self, d0=L.tgt.data_ptr(), ldd0=256,
        d1=L.tgt_id.data_ptr(), ldd1=256, csplit=256, accumulate=True, nsplit=ns), args.iters)
    for tile, ks in ((0, 4), (64, 4), (192, 4), (0, 2), (0, 8)):
        parts = torch.zeros(ks, N, 512, device=dev)
        res[f"proj_ls_splitk(tile{tile},ks{ks})"] = timeit(lambda: hip.linear(
            L.Ylt, W.Wp_ls, N, 512, 2048, ldx=1024, ldy=2048, x2=L.Yst, ldx2=1024, kx_split=1024, bias=W.bp_ls,
            nsplit=ns, tile=tile, ksplits=ks, parts=parts, part_stride=N * 512), args.iters)
        res[f"proj_self_splitk(tile{tile},ks{ks})"] = timeit(lambda: hip.linear(
            L.Ylt, W.Wp_self, N, 512, 1024, ldx=1024, ldy=1024, bias=W.bp_self, nsplit=ns, tile=tile, ksplits=ks,
            parts=parts, part_stride=N * 512), args.iters)
    Ucat = L.Ucat
    grp = lambda: hip.linear_grouped([
        hip.linear(L.x_pl, W.Wq, N, 128, 256, ldx=256, ldy=256, bias=W.bq, pa=curK,
                   ldpa=128, pb=L.Qpe, ldpb=128, addvec=L.cur_pe, nsplit=ns, tile=64, launch=False),
        hip.linear(L.x_pl, W.Wrel_x, N, 225, 256, ldx=256, ldy=256, bias=W.brel_x, d0=L.R.data_ptr(), ldd0=L.ldr,
                   d0_cs=L.rcs, nsplit=ns, tile=64, launch=False),
        hip.linear(L.x_pl, W.pe_x[T][0], N, T, 256, ldx=256, ldy=256, bias=W.pe_x[T][1], d0=L.bias_pe.data_ptr(),
                   ldd0=T, nsplit=ns, tile=64, launch=False),
        hip.linear(L.x_pl, W.Wv, N, 512, 256, ldx=256, ldy=256, bias=W.bv, act=1, pa=curV, ldpa=1024,
                   pa_blocked=True, nsplit=ns, tile=64, launch=False),
        hip.linear(L.x_pl, W.Wu, N, 512, 256, ldx=256, ldy=256, bias=W.bu, act=1, d0=Ucat.data_ptr(), ldd0=1024,
                   nsplit=ns, tile=64, launch=False),
        hip.linear(L.z_pl[1], W.Widu, N, 512, 256, ldx=256, ldy=256, bias=W.bidu, act=1,
                   d0=Ucat.data_ptr() + 512 * 4, ldd0=1024, nsplit=ns, tile=64, launch=False)])
    res["grouped_Q_R_pe_V_U_IDU"] = timeit(grp, args.iters)
    res["gemm_R(225x128)"] = timeit(lambda: hip.linear(curK, W.Wrel, N, 225, 128, ldx=128, ldy=128, bias=W.brel,
                                                       d0=L.R.data_ptr(), ldd0=L.ldr, d0_cs=L.rcs, nsplit=ns), args.iters)
    lab = torch.randint(0, 11, (481, 849), dtype=torch.uint8, device=dev) if (args.h, args.w) == (31, 54) else \
        torch.randint(0, 11, ((args.h - 1) * 16 + 1, (args.w - 1) * 16 + 1), dtype=torch.uint8, device=dev)
    res["id_assign"] = timeit(lambda: L.assign_identity(lab), args.iters)
    emb = torch.randn(N, 256, device=dev)
    res["lstt_forward_frame"] = timeit(lambda: L.forward(emb), max(5, args.iters // 3))
    L.cur = T
    print(json.dumps({k: round(v, 2) for k, v in res.items()}, indent=1))


if __name__ == "__main__":
    main()
