#!/usr/bin/env python
"""The four projection launches of a GPM layer exactly as rmem_amd/lstt.py issues them (grouped front launch, split-K
projection of the two memory reads, grouped self-attention launch, split-K self projection), each timed as 20 launches
inside one hipGraph (tools/kbench.timeit), plus the same split-K projections with every operand's leading dimension padded
by 64 elements (rows no longer a power-of-two number of bytes apart: an L2-channel probe).

    python tools/kbench_proj.py            # timings (JSON)
    python tools/kbench_proj.py --trace    # cycle stamps of the four launches (rmem_linear_trace)"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kbench import timeit  # noqa: E402


def build():
    from rmem_amd import hip
    from rmem_amd.config import get_config
    from rmem_amd.lstt import DeAOTLSTT
    from rmem_amd.model import build_vos_model
    from rmem_amd.synth import load_synthetic_weights
    dev = torch.device("cuda:0")
    cfg = get_config("r50_deaotl", 1, 3)
    model = build_vos_model("deaot", cfg).eval()
    load_synthetic_weights(model)
    model = model.to(dev)
    L = DeAOTLSTT(model, 31, 54, dev, nsplit=3)
    return hip, L, dev


def problems(hip, L, l=1, T=4):
    """name -> list of LinearArgs (one launch each), as _forward_layer builds them for layer l."""
    N, ns = L.N, L.nsplit
    W = L.lw[l]
    L.tgt.normal_()
    L._ln(L.tgt, W.ln1, L.x_pl, 256)
    L._ln(L.tgt, W.ln1, L.z_pl[l], 256)
    L._ln(L.tgt, W.ln1, L.s_pl, 512)
    curK, curV, Ucat = L.bankK[l][T], L.bankV[l][T], L.Ucat
    pe = W.pe_x[T]
    front = [
        hip.linear(L.x_pl, W.Wq, N, 128, 256, ldx=256, ldy=256, bias=W.bq, pa=curK, ldpa=128, pb=L.Qpe, ldpb=128,
                   addvec=L.cur_pe, nsplit=ns, launch=False),
        hip.linear(L.x_pl, W.Wrel_x, N, L.WIN, 256, ldx=256, ldy=256, bias=W.brel_x, d0=L.R.data_ptr(), ldd0=L.ldr,
                   d0_cs=L.rcs, nsplit=ns, launch=False),
        hip.linear(L.x_pl, pe[0], N, T, 256, ldx=256, ldy=256, bias=pe[1], d0=L.bias_pe.data_ptr(), ldd0=T, nsplit=ns,
                   launch=False),
        hip.linear(L.x_pl, W.Wv, N, 512, 256, ldx=256, ldy=256, bias=W.bv, act=1, pa=curV, ldpa=1024, pa_blocked=True,
                   nsplit=ns, launch=False),
        hip.linear(L.x_pl, W.Wu, N, 512, 256, ldx=256, ldy=256, bias=W.bu, act=1, d0=Ucat.data_ptr(), ldd0=1024, nsplit=ns,
                   launch=False),
        hip.linear(L.z_pl[l], W.Widu, N, 512, 256, ldx=256, ldy=256, bias=W.bidu, act=1, d0=Ucat.data_ptr() + 512 * 4,
                   ldd0=1024, nsplit=ns, launch=False)]
    proj_ls = [hip.linear(L.Ylt, W.Wp_ls, N, 512, 2048, ldx=1024, ldy=2048, x2=L.Yst, ldx2=1024, kx_split=1024, bias=W.bp_ls,
                          nsplit=ns, ksplits=L.KS, parts=L.parts, part_stride=N * 512, launch=False)]
    sQK = hip.Planes(L.selfQK.hi[0], L.selfQK.lo[0])
    self_front = [
        hip.linear(L.s_pl, W.Wqk, N, 128, 512, ldx=512, ldy=512, bias=W.bqk, pa=sQK, ldpa=128, nsplit=ns, launch=False),
        hip.linear(L.s_pl, W.Wv12, N, 512, 256, ldx=512, ldy=256, bias=W.bv12, act=1, pa=L.selfV, ldpa=1024, pa_blocked=True,
                   nbatch=2, bsx=256, bsy=512 * 256, bsbias=512, bspa=512 * 16, nsplit=ns, launch=False),
        hip.linear(L.s_pl, W.Wu12, N, 512, 256, ldx=512, ldy=256, bias=W.bu12, act=1, d0=L.Uself.data_ptr(), ldd0=1024, nbatch=2,
                   bsx=256, bsy=512 * 256, bsbias=512, bsd=512, nsplit=ns, launch=False)]
    proj_self = [hip.linear(L.Ylt, W.Wp_self, N, 512, 1024, ldx=1024, ldy=1024, bias=W.bp_self, nsplit=ns, ksplits=L.KS,
                            parts=L.parts, part_stride=N * 512, launch=False)]
    return {"front": front, "proj_ls": proj_ls, "self_front": self_front, "proj_self": proj_self}


def padded(hip, pl, pad=64):
    """Copy of planes [rows][K] with the leading dimension K + pad."""
    rows, K = pl.hi.shape[-2], pl.hi.shape[-1]
    hi = torch.zeros(rows, K + pad, dtype=pl.hi.dtype, device=pl.hi.device)
    lo = torch.zeros_like(hi)
    hi[:, :K], lo[:, :K] = pl.hi.reshape(rows, K), pl.lo.reshape(rows, K)
    return hip.Planes(hi, lo), K + pad


def main():
    hip, L, dev = build()
    P = problems(hip, L)
    res = {}
    for name, grp in P.items():
        res[name] = timeit(lambda: hip.linear_grouped(grp), 20)
    res["sum4"] = sum(res[k] for k in P)
    # the split-K projections with padded leading dimensions
    N, ns, W = L.N, L.nsplit, L.lw[1]
    Xa, lda = padded(hip, L.Ylt)
    Xb, ldb = padded(hip, L.Yst)
    Wls, ldw = padded(hip, W.Wp_ls)
    Wse, ldw2 = padded(hip, W.Wp_self)
    keep = [Xa, Xb, Wls, Wse]
    a = hip.linear(Xa, Wls, N, 512, 2048, ldx=lda, ldy=ldw, x2=Xb, ldx2=ldb, kx_split=1024, bias=W.bp_ls, nsplit=ns,
                   ksplits=L.KS, parts=L.parts, part_stride=N * 512, launch=False)
    res["proj_ls_padded_ld"] = timeit(lambda: hip.linear_grouped([a]), 20)
    b = hip.linear(Xa, Wse, N, 512, 1024, ldx=lda, ldy=ldw2, bias=W.bp_self, nsplit=ns, ksplits=L.KS, parts=L.parts,
                   part_stride=N * 512, launch=False)
    res["proj_self_padded_ld"] = timeit(lambda: hip.linear_grouped([b]), 20)
    del keep
    print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in res.items()}))


def trace_main():
    hip, L, dev = build()
    P = problems(hip, L)
    lib = hip.load()
    if os.environ.get("RMEM_STREAM_VAR"):      # timing experiments of the tracing kernel: 2 no requests, 3 no MFMAs, 4 no fragment reads
        hip.configure("stream_var", int(os.environ["RMEM_STREAM_VAR"]))
    out = {}
    for name, grp in P.items():
        arr = (hip.LinearArgs * len(grp))(*grp)
        tr = torch.zeros(256, 64, dtype=torch.int64, device=dev)
        for _ in range(3):
            tr.zero_()
            hip.check(lib.rmem_linear_trace(arr, len(grp), tr.data_ptr(), hip.stream_ptr()), "trace")
        torch.cuda.synchronize()
        t = tr.cpu()
        rows = []
        ends = []
        rt0 = [int(t[b][60]) for b in range(256) if int(t[b][63]) != 0]
        rt1 = [int(t[b][61]) for b in range(256) if int(t[b][63]) != 0]
        for b in range(256):
            r = t[b]
            if int(r[63]) == 0:
                continue
            ends.append(int(r[63] - r[0]))
            if b in (0, 1, 100, 255):
                n = int(r[62])
                stamps = [int(r[1] - r[0])] + [int(r[2 + i] - r[0]) for i in range(min(2 * n, 60))] + [int(r[63] - r[0])]
                rows.append({"block": b, "stages": n, "cycles_since_start": stamps})
        ends.sort()
        # device-wide 100 MHz clock: first start -> last end of the launch, spread of the starts, and the shader clock implied
        span = (max(rt1) - min(rt0)) / 100.0
        durs = sorted((b - a) / 100.0 for a, b in zip(rt0, rt1))
        out[name] = {"workgroups": len(ends), "end_cycles_median": ends[len(ends) // 2], "end_cycles_max": ends[-1],
                     "span_us": round(span, 2), "start_spread_us": round((max(rt0) - min(rt0)) / 100.0, 2),
                     "wg_us_median": round(durs[len(durs) // 2], 2), "wg_us_max": round(durs[-1], 2),
                     "ghz": round(ends[len(ends) // 2] / max(durs[len(durs) // 2], 1e-9) / 1e3, 3), "rows": rows}
    print(json.dumps(out))


if __name__ == "__main__":
    if "--trace" in sys.argv:
        trace_main()
    else:
        main()
