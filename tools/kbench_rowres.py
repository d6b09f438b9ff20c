#!/usr/bin/env python
"""The fused LayerNorm + grouped projection launch (rmem_ln_linear_grouped, csrc/linear_rowres.h) against the launches it
replaces (rmem_layernorm_red2 + rmem_linear_grouped), at 480p, for the two launch shapes of a GPM layer -- each timed as 20
launches inside one hipGraph -- and, with --trace, the per-wave cycle stamps of the fused kernel (median / max over the
waves of the launch, cycles since the wave's start)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kbench import timeit  # noqa: E402


def main():
    from rmem_amd import hip
    from rmem_amd.config import get_config
    from rmem_amd.lstt import DeAOTLSTT
    from rmem_amd.model import build_vos_model
    from rmem_amd.synth import load_synthetic_weights
    from rmem_amd.hip import Planes
    os.environ["RMEM_ROWRES"] = "fused"      # (allocates the second pair of residual streams)
    dev = torch.device("cuda:0")
    cfg = get_config("r50_deaotl", 1, 3)
    model = build_vos_model("deaot", cfg).eval()
    load_synthetic_weights(model)
    model = model.to(dev)
    h, w = (46, 81) if "--720p" in sys.argv else (31, 54)
    L = DeAOTLSTT(model, h, w, dev, nsplit=3)
    N, T, ns = L.N, 4, 3
    W = L.lw[1]
    L.tgt.normal_()
    L.tgt_id.normal_()
    L.parts.normal_()
    curK, curV, Ucat = L.bankK[1][T], L.bankV[1][T], L.Ucat
    pe = W.pe_x[T]
    front = [
        hip.linear(L.x_pl, W.Wq, N, 128, 256, ldx=256, ldy=256, bias=W.bq, pa=curK, ldpa=128, pb=L.Qpe, ldpb=128,
                   addvec=L.cur_pe, nsplit=ns, launch=False),
        hip.linear(L.x_pl, W.Wrel_x, N, 225, 256, ldx=256, ldy=256, bias=W.brel_x, d0=L.R.data_ptr(), ldd0=L.ldr,
                   d0_cs=L.rcs, nsplit=ns, launch=False),
        hip.linear(L.x_pl, pe[0], N, T, 256, ldx=256, ldy=256, bias=pe[1], d0=L.bias_pe.data_ptr(), ldd0=T, nsplit=ns,
                   launch=False),
        hip.linear(L.x_pl, W.Wv, N, 512, 256, ldx=256, ldy=256, bias=W.bv, act=1, pa=curV, ldpa=1024, pa_blocked=True,
                   nsplit=ns, launch=False),
        hip.linear(L.x_pl, W.Wu, N, 512, 256, ldx=256, ldy=256, bias=W.bu, act=1, d0=Ucat.data_ptr(), ldd0=1024, nsplit=ns,
                   launch=False),
        hip.linear(L.z_pl[1], W.Widu, N, 512, 256, ldx=256, ldy=256, bias=W.bidu, act=1, d0=Ucat.data_ptr() + 512 * 4,
                   ldd0=1024, nsplit=ns, launch=False)]
    sQK = Planes(L.selfQK.hi[0], L.selfQK.lo[0])
    selfg = [
        hip.linear(L.s_pl, W.Wqk, N, 128, 512, ldx=512, ldy=512, bias=W.bqk, pa=sQK, ldpa=128, nsplit=ns, launch=False),
        hip.linear(L.s_pl, W.Wv12, N, 512, 256, ldx=512, ldy=256, bias=W.bv12, act=1, pa=L.selfV, ldpa=1024,
                   pa_blocked=True, nbatch=2, bsx=256, bsy=512 * 256, bsbias=512, bspa=512 * 16, nsplit=ns, launch=False),
        hip.linear(L.s_pl, W.Wu12, N, 512, 256, ldx=512, ldy=256, bias=W.bu12, act=1, d0=L.Uself.data_ptr(), ldd0=1024,
                   nbatch=2, bsx=256, bsy=512 * 256, bsbias=512, bsd=512, nsplit=ns, launch=False)]
    fr = [W.Wq_f, W.Wrel_f, W.pe_f[T], W.Wv_f, W.Wu_f, W.Widu_f]
    pp = L.parts.data_ptr()

    def fused_front(trace=None, nparts=L.KS):
        hip.ln_linear_grouped(
            [hip.rowres_stream(x=L.tgt, xo=L.tgt_b, parts=pp, gamma=W.ln1[0], beta=W.ln1[1]),
             hip.rowres_stream(x=L.tgt_id, xo=L.tgt_id_b, parts=pp + 1024, gamma=W.lnid1[0], beta=W.lnid1[1],
                               planes=L.z_pl[1], ldo=256)],
            0, N, nparts, N * 512, 512, 1e-5, [(a, f, 256 if i == 5 else 0, 0) for i, (a, f) in enumerate(zip(front, fr))],
            trace=trace)

    def fused_front_planes(trace=None):
        hip.ln_linear_grouped([hip.rowres_stream(planes=L.x_pl, ldo=256)], 1, N, 0, 0, 0, 1e-5,
                              [(a, f, 0, 0) for a, f in zip(front[:5], fr[:5])], trace=trace)

    def fused_self(trace=None):
        hip.ln_linear_grouped(
            [hip.rowres_stream(x=L.tgt, xo=L.tgt_b, parts=pp, gamma=W.ln2[0], beta=W.ln2[1]),
             hip.rowres_stream(x=L.tgt_id, xo=L.tgt_id_b, parts=pp + 1024, gamma=W.lnid2[0], beta=W.lnid2[1])],
            0, N, L.KS, N * 512, 512, 1e-5,
            [(selfg[0], W.Wqk_f, 0, 0), (selfg[1], W.Wv12_f, 0, 256), (selfg[2], W.Wu12_f, 0, 256)], trace=trace)

    def old_front():
        L._ln2(W.ln1, L.x_pl, 256, 0, W.lnid1, L.z_pl[1], 256, 0, parts=True)
        hip.linear_grouped(front)

    def old_self():
        L._ln2(W.ln2, L.s_pl, 512, 0, W.lnid2, L.s_pl, 512, 256, parts=True)
        hip.linear_grouped(selfg)

    res = {"N": N}
    res["old_front_ln+grouped"] = timeit(old_front, 20)
    res["old_front_grouped_only"] = timeit(lambda: hip.linear_grouped(front), 20)
    res["fused_front"] = timeit(fused_front, 20)
    res["fused_front_nparts0"] = timeit(lambda: fused_front(nparts=0), 20)
    res["fused_front_planes_mode1"] = timeit(fused_front_planes, 20)
    def planes_self():
        L._ln2(W.ln2, L.s_pl, 512, 0, W.lnid2, L.s_pl, 512, 256, parts=True)
        hip.ln_linear_grouped([hip.rowres_stream(planes=L.s_pl, ldo=512), hip.rowres_stream(planes=L.s_pl, ldo=512, plane_off=256)],
                              1, N, 0, 0, 0, 1e-5, [(selfg[0], W.Wqk_f, 0, 0), (selfg[1], W.Wv12_f, 0, 256), (selfg[2], W.Wu12_f, 0, 256)])

    def planes_front():
        L._ln2(W.ln1, L.x_pl, 256, 0, W.lnid1, L.z_pl[1], 256, 0, parts=True)
        hip.ln_linear_grouped([hip.rowres_stream(planes=L.x_pl, ldo=256), hip.rowres_stream(planes=L.z_pl[1], ldo=256)],
                              1, N, 0, 0, 0, 1e-5, [(a, f, 256 if i == 5 else 0, 0) for i, (a, f) in enumerate(zip(front, fr))])

    res["ln2_only"] = timeit(lambda: L._ln2(W.ln1, L.x_pl, 256, 0, W.lnid1, L.z_pl[1], 256, 0, parts=True), 20)
    res["planes_front_ln+rowres"] = timeit(planes_front, 20)
    res["planes_self_ln+rowres"] = timeit(planes_self, 20)
    res["old_self_ln+grouped"] = timeit(old_self, 20)
    res["fused_self"] = timeit(fused_self, 20)
    print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in res.items()}))
    if "--trace" in sys.argv:
        out = {}
        for name, fn in (("front", fused_front), ("front_planes", fused_front_planes), ("self", fused_self)):
            tr = torch.zeros(8 * ((N + 511) // 512) * 16 * 64, dtype=torch.int64, device=dev)
            for _ in range(3):
                tr.zero_()
                fn(trace=tr)
            torch.cuda.synchronize()
            t = tr.view(-1, 8).cpu()
            t = t[t[:, 5] > 0]                              # waves that ran a unit
            d = (t[:, 1:6] - t[:, 0:1]).float()
            span = int(t[:, 5].max() - t[:, 0].min())
            out[name] = {"waves": int(t.shape[0]),
                         "median_cycles_[w_issued,ln_done,barrier,mfma_done,end]": [int(x) for x in d.median(0).values],
                         "max": [int(x) for x in d.max(0).values], "first_start_to_last_end": span}
        print(json.dumps(out))


if __name__ == "__main__":
    main()
