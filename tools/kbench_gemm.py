#!/usr/bin/env python
"""Where the time of the small projection GEMMs goes: the grouped front launch of a layer (Q, relative bias,
temporal-PE bias, V, U, ID_U) and the split-K projections, whole and in parts, at K = 64 / 128 / 256 -- each variant
timed as 20 launches inside one hipGraph (no Python launch cost in the number)."""
import json

TILE, TILE2 = 0, 0
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kbench import timeit  # noqa: E402


def main():
    global TILE, TILE2
    stream = "--tiles" not in sys.argv          # default: the streaming kernel (tile 0); --tiles: the tile-per-workgroup kernels
    TILE, TILE2 = (0, 0) if stream else (64, 192)
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
    N, T, ns = L.N, 4, 3
    W = L.lw[1]
    L.tgt.normal_()
    L._ln(L.tgt, W.ln1, L.x_pl, 256)
    L._ln(L.tgt, W.ln1, L.z_pl[1], 256)
    curK, curV, Ucat = L.bankK[1][T], L.bankV[1][T], L.Ucat
    res = {"kernels": "stream" if stream else "tiles"}

    def members(K):
        return {
            "Q": hip.linear(L.x_pl, W.Wq, N, 128, K, ldx=256, ldy=256, bias=W.bq, pa=curK, ldpa=128, pb=L.Qpe, ldpb=128,
                            addvec=L.cur_pe, nsplit=ns, tile=TILE, launch=False),
            "R": hip.linear(L.x_pl, W.Wrel_x, N, 225, K, ldx=256, ldy=256, bias=W.brel_x, d0=L.R.data_ptr(), ldd0=L.ldr,
                            d0_cs=L.rcs, nsplit=ns, tile=TILE, launch=False),
            "Rplain": hip.linear(L.x_pl, W.Wrel_x, N, 225, K, ldx=256, ldy=256, bias=W.brel_x, d0=L.R.data_ptr(), ldd0=225,
                                 nsplit=ns, tile=TILE, launch=False),
            "pe": hip.linear(L.x_pl, W.pe_x[T][0], N, T, K, ldx=256, ldy=256, bias=W.pe_x[T][1], d0=L.bias_pe.data_ptr(),
                             ldd0=T, nsplit=ns, tile=TILE, launch=False),
            "V": hip.linear(L.x_pl, W.Wv, N, 512, K, ldx=256, ldy=256, bias=W.bv, act=1, pa=curV, ldpa=1024,
                            pa_blocked=True, nsplit=ns, tile=TILE, launch=False),
            "U": hip.linear(L.x_pl, W.Wu, N, 512, K, ldx=256, ldy=256, bias=W.bu, act=1, d0=Ucat.data_ptr(), ldd0=1024,
                            nsplit=ns, tile=TILE, launch=False),
            "IDU": hip.linear(L.z_pl[1], W.Widu, N, 512, K, ldx=256, ldy=256, bias=W.bidu, act=1,
                              d0=Ucat.data_ptr() + 512 * 4, ldd0=1024, nsplit=ns, tile=TILE, launch=False)}

    for K in (256, 128, 64):
        m = members(K)
        res[f"front_all_K{K}"] = timeit(lambda: hip.linear_grouped([m[k] for k in ("Q", "R", "pe", "V", "U", "IDU")]), 20)
    m = members(256)
    for name in ("Q", "R", "Rplain", "pe", "V", "U", "IDU"):
        res[f"only_{name}"] = timeit(lambda: hip.linear_grouped([m[name]]), 20)
    res["V+U"] = timeit(lambda: hip.linear_grouped([m["V"], m["U"]]), 20)
    res["V+U+IDU"] = timeit(lambda: hip.linear_grouped([m["V"], m["U"], m["IDU"]]), 20)
    res["all_but_R"] = timeit(lambda: hip.linear_grouped([m[k] for k in ("Q", "pe", "V", "U", "IDU")]), 20)
    # the split-K projections, whole and at a quarter of the depth (own partial buffer: L.parts holds L.KS = 2 splits)
    parts4 = torch.zeros(4, N, 512, device=dev)
    for Kp in (2048, 512):
        res[f"proj_ls_K{Kp}_tile192_ks4"] = timeit(lambda: hip.linear(
            L.Ylt, W.Wp_ls, N, 512, Kp, ldx=1024, ldy=2048, x2=L.Yst, ldx2=1024, kx_split=min(1024, Kp // 2), bias=W.bp_ls,
            nsplit=ns, tile=TILE2, ksplits=4, parts=parts4, part_stride=N * 512), 20)
    res["proj_self_K1024_tile192_ks4"] = timeit(lambda: hip.linear(
        L.Ylt, W.Wp_self, N, 512, 1024, ldx=1024, ldy=1024, bias=W.bp_self, nsplit=ns, tile=TILE2, ksplits=4,
        parts=parts4, part_stride=N * 512), 20)
    res["proj_self_K1024_direct_tile64"] = timeit(lambda: hip.linear(
        L.Ylt, W.Wp_self, N, 512, 1024, ldx=1024, ldy=1024, bias=W.bp_self, d0=L.tgt.data_ptr(), ldd0=256,
        d1=L.tgt_id.data_ptr(), ldd1=256, csplit=256, accumulate=True, nsplit=ns), 20)
    res["ln2"] = timeit(lambda: L._ln2(W.ln2, L.s_pl, 512, 0, W.lnid2, L.s_pl, 512, 256, parts=True), 20)
    res["dwconv_1map"] = timeit(lambda: L._dwconv(L.ws_main, W.dw_lt, L.Ylt), 20)
    lib = hip.load()
    res["dwconv_2maps"] = timeit(lambda: hip.check(lib.rmem_dwconv5x5_split2(
        L.ws_main.G.data_ptr(), L.ws_side.G.data_ptr(), 1024, W.dw_lt.data_ptr(), W.dw_st.data_ptr(), L.h, L.w, 1024,
        L.Ylt.hi.data_ptr(), L.Ylt.lo.data_ptr(), L.Yst.hi.data_ptr(), L.Yst.lo.data_ptr(), 1024, hip.stream_ptr()), "dw2"), 20)
    res["set_ints(empty-ish kernel)"] = timeit(lambda: hip.set_ints(L.maps, [0, 1], offset=16, count=2), 20)
    print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in res.items()}, indent=1))


if __name__ == "__main__" and "--trace" not in sys.argv:
    main()


def trace_main():
    """--trace: cycle stamps of the streaming kernel (rmem_linear_trace) for the 4-column problem alone (27 items, one per
    workgroup) and for the whole grouped front launch."""
    import ctypes as C
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
    N, T, ns = L.N, 4, 3
    W = L.lw[1]
    L.tgt.normal_()
    L._ln(L.tgt, W.ln1, L.x_pl, 256)
    L._ln(L.tgt, W.ln1, L.z_pl[1], 256)
    curK, curV, Ucat = L.bankK[1][T], L.bankV[1][T], L.Ucat
    m = {
        "Q": hip.linear(L.x_pl, W.Wq, N, 128, 256, ldx=256, ldy=256, bias=W.bq, pa=curK, ldpa=128, pb=L.Qpe, ldpb=128,
                        addvec=L.cur_pe, nsplit=ns, launch=False),
        "R": hip.linear(L.x_pl, W.Wrel_x, N, 225, 256, ldx=256, ldy=256, bias=W.brel_x, d0=L.R.data_ptr(), ldd0=L.ldr,
                        d0_cs=L.rcs, nsplit=ns, launch=False),
        "pe": hip.linear(L.x_pl, W.pe_x[T][0], N, T, 256, ldx=256, ldy=256, bias=W.pe_x[T][1], d0=L.bias_pe.data_ptr(),
                         ldd0=T, nsplit=ns, launch=False),
        "V": hip.linear(L.x_pl, W.Wv, N, 512, 256, ldx=256, ldy=256, bias=W.bv, act=1, pa=curV, ldpa=1024,
                        pa_blocked=True, nsplit=ns, launch=False),
        "U": hip.linear(L.x_pl, W.Wu, N, 512, 256, ldx=256, ldy=256, bias=W.bu, act=1, d0=Ucat.data_ptr(), ldd0=1024,
                        nsplit=ns, launch=False),
        "IDU": hip.linear(L.z_pl[1], W.Widu, N, 512, 256, ldx=256, ldy=256, bias=W.bidu, act=1,
                          d0=Ucat.data_ptr() + 512 * 4, ldd0=1024, nsplit=ns, launch=False)}
    lib = hip.load()
    out = {}
    for name, keys in (("only_pe", ["pe"]), ("only_V", ["V"]), ("front_all", ["Q", "R", "pe", "V", "U", "IDU"])):
        arr = (hip.LinearArgs * len(keys))(*[m[k] for k in keys])
        tr = torch.zeros(256, 64, dtype=torch.int64, device=dev)
        for _ in range(3):
            hip.check(lib.rmem_linear_trace(arr, len(keys), tr.data_ptr(), hip.stream_ptr()), "trace")
        torch.cuda.synchronize()
        t = tr.cpu()
        rows = []
        for b in (0, 1, 100, 255):
            r = t[b]
            if int(r[63]) == 0:
                continue
            n = int(r[62])
            stamps = [int(r[1] - r[0])] + [int(r[2 + i] - r[0]) for i in range(min(2 * n, 60))] + [int(r[63] - r[0])]
            rows.append({"block": b, "stages": n, "cycles_since_start": stamps})
        out[name] = rows
    print(json.dumps(out))


if __name__ == "__main__" and "--trace" in sys.argv:
    trace_main()
