"""GPU: every C-ABI kernel of librmem_hip.so against an fp64 / oracle reference on
seeded inputs (op-level parity).  All calls go through the C ABI
(rmem_amd.hip ctypes binding)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def hip():
    from rmem_amd import hip as H
    H.load()
    return H


def _rand(rs, *shape, scale=1.0):
    return torch.from_numpy(rs.standard_normal(shape).astype(np.float32) * np.float32(scale))


def _planes(H, x):
    return H.Planes.from_f32(x.to(DEV).contiguous())


def _silu(x):
    return x * torch.sigmoid(x)


# tolerance for nsplit=3 (fp32-class) and nsplit=1 (plain bf16) relative to max|ref|
def _tol(nsplit):
    return 3e-5 if nsplit == 3 else 2e-2


@pytest.mark.parametrize("nsplit", [3, 1])
@pytest.mark.parametrize("tile", [0, 64, 192])
@pytest.mark.parametrize("M,N,K", [(200, 130, 256), (64, 64, 64), (333, 225, 128)])
def test_linear_basic(hip, nsplit, tile, M, N, K):
    rs = np.random.RandomState(M + N + K)
    X, Y, b = _rand(rs, M, K), _rand(rs, N, K), _rand(rs, N)
    ref = (X.double() @ Y.double().t() + b.double())
    d = torch.full((M, N + 3), 7.0, device=DEV)
    pa = hip.Planes.empty((M, N + 5), DEV)
    pb = hip.Planes.empty((M, N), DEV)
    addv = _rand(rs, N).to(DEV)
    hip.linear(_planes(hip, X), _planes(hip, Y), M, N, K, ldx=K, ldy=K, bias=b.to(DEV),
               d0=d.data_ptr(), ldd0=N + 3, pa=pa, ldpa=N + 5, pb=pb, ldpb=N, addvec=addv,
               nsplit=nsplit, tile=tile)
    torch.cuda.synchronize()
    out = d[:, :N].cpu().double()
    scale = ref.abs().max().item()
    assert (out - ref).abs().max().item() < _tol(nsplit) * scale
    assert torch.all(d[:, N:] == 7.0)                      # no writes outside [M][N]
    assert (pa.float()[:, :N].cpu().double() - ref).abs().max().item() < (_tol(nsplit) + 2e-5) * scale
    refb = ref + addv.cpu().double()
    assert (pb.float().cpu().double() - refb).abs().max().item() < (_tol(nsplit) + 2e-5) * refb.abs().max().item()


def test_linear_swapped_silu_batch_segments(hip):
    """bias per row + SiLU + nbatch=2 + two K segments + accumulate/csplit."""
    rs = np.random.RandomState(5)
    # swapped, batched (self-attn V^T pattern): D_b[512][Ntok] = silu(W_b . S[:, 256b:256b+256]^T + bias_b)
    ntok, Np = 150, 256
    Wt, S, b = _rand(rs, 1024, 256, scale=0.1), _rand(rs, ntok, 512), _rand(rs, 1024)
    pa = hip.Planes.empty((1024, Np), DEV)
    hip.linear(_planes(hip, Wt), _planes(hip, S), 512, ntok, 256, ldx=256, ldy=512, bias=b.to(DEV),
               bias_per_row=True, act=1, pa=pa, ldpa=Np, nbatch=2, bsx=512 * 256, bsy=256, bsbias=512,
               bspa=512 * Np)
    torch.cuda.synchronize()
    ref = torch.cat([_silu(Wt[:512].double() @ S[:, :256].double().t() + b[:512, None].double()),
                     _silu(Wt[512:].double() @ S[:, 256:].double().t() + b[512:, None].double())], 0)
    got = pa.float()[:, :ntok].cpu().double()
    assert (got - ref).abs().max().item() < 5e-5 * ref.abs().max().item()
    assert torch.all(pa.hi[:, ntok:] == 0)

    # two K segments on X, accumulate into two destinations split at column 256
    M, N = 170, 512
    X1, X2, Wp, bias = _rand(rs, M, 1024), _rand(rs, M, 1024), _rand(rs, N, 2048, scale=0.05), _rand(rs, N)
    t0, t1 = _rand(rs, M, 256).to(DEV), _rand(rs, M, 256).to(DEV)
    r0, r1 = t0.clone(), t1.clone()
    hip.linear(_planes(hip, X1), _planes(hip, Wp), M, N, 2048, ldx=1024, ldy=2048, x2=_planes(hip, X2),
               ldx2=1024, kx_split=1024, bias=bias.to(DEV), d0=t0.data_ptr(), ldd0=256,
               d1=t1.data_ptr(), ldd1=256, csplit=256, accumulate=True)
    torch.cuda.synchronize()
    ref = torch.cat([X1, X2], 1).double() @ Wp.double().t() + bias.double()
    assert (t0.cpu().double() - (r0.cpu().double() + ref[:, :256])).abs().max().item() < 1e-4
    assert (t1.cpu().double() - (r1.cpu().double() + ref[:, 256:])).abs().max().item() < 1e-4

    # two K segments on Y (ID_V pattern): D[512][ntok] = W[512][512] . [Z | E]^T
    Wi, Z, E = _rand(rs, 512, 512, scale=0.1), _rand(rs, ntok, 256), _rand(rs, ntok, 256)
    pa = hip.Planes.empty((512, Np), DEV)
    hip.linear(_planes(hip, Wi), _planes(hip, Z), 512, ntok, 512, ldx=512, ldy=256, y2=_planes(hip, E),
               ldy2=256, ky_split=256, pa=pa, ldpa=Np)
    torch.cuda.synchronize()
    ref = Wi.double() @ torch.cat([Z, E], 1).double().t()
    assert (pa.float()[:, :ntok].cpu().double() - ref).abs().max().item() < 5e-5 * ref.abs().max().item()


def test_layernorm_dwconv_groupnorm(hip):
    from oracle import lstt_ref as R
    lib, st = hip.load(), hip.stream_ptr()
    rs = np.random.RandomState(3)
    h, w = 7, 9
    N = h * w
    g_ = lambda t: t.to(DEV).contiguous()      # keep device tensors alive across the launch
    # layernorm -> planes (+ fp32)
    x, g, b = _rand(rs, N, 256, scale=2.0) + 0.5, _rand(rs, 256) * 0.2 + 1, _rand(rs, 256) * 0.1
    dx, dg, db = g_(x), g_(g), g_(b)
    pl = hip.Planes.empty((N, 512), DEV)
    of = torch.zeros(N, 256, device=DEV)
    hip.check(lib.rmem_layernorm_split(dx.data_ptr(), 256, dg.data_ptr(), db.data_ptr(),
                                       N, 256, 1e-5, pl.hi.data_ptr() + 256 * 2, pl.lo.data_ptr() + 256 * 2,
                                       512, of.data_ptr(), 256, st), "ln")
    ref = R.layer_norm(x, g, b)
    assert (of.cpu() - ref).abs().max().item() < 2e-6
    assert (pl.float()[:, 256:].cpu() - ref).abs().max().item() < 3e-5
    assert torch.all(pl.hi[:, :256] == 0)
    # depth-wise conv
    gin, wdw = _rand(rs, N, 1024), _rand(rs, 1024, 1, 5, 5, scale=0.25)
    dgin = g_(gin)
    pl = hip.Planes.empty((N, 1024), DEV)
    wt = wdw.reshape(1024, 25).t().contiguous().to(DEV)
    hip.check(lib.rmem_dwconv5x5_split(dgin.data_ptr(), 1024, wt.data_ptr(), h, w, 1024,
                                       pl.hi.data_ptr(), pl.lo.data_ptr(), 1024, st), "dwconv")
    ref = R.dwconv5x5(gin, wdw, h, w)
    assert (pl.float().cpu() - ref).abs().max().item() < 3e-5
    # group norm (2 groups)
    t0, t1 = _rand(rs, N, 256, scale=1.5) + 0.3, _rand(rs, N, 256, scale=0.7) - 0.2
    gg, gb = _rand(rs, 512) * 0.2 + 1, _rand(rs, 512) * 0.1
    dt0, dt1, dgg, dgb = g_(t0), g_(t1), g_(gg), g_(gb)
    ws = torch.zeros(4 * ((N + 63) // 64), dtype=torch.float64, device=DEV)
    o = torch.zeros(N, 512, device=DEV)
    hip.check(lib.rmem_groupnorm2(dt0.data_ptr(), dt1.data_ptr(), N, 256, dgg.data_ptr(),
                                  dgb.data_ptr(), 1e-5, ws.data_ptr(), o.data_ptr(), 512, st), "gn")
    ref = R.group_norm_tokens(torch.cat([t0, t1], 1), gg, gb, 2)
    assert (o.cpu() - ref).abs().max().item() < 5e-6
    # mass reduce
    mass, fg = torch.rand(N, 3), torch.rand(N)
    dmass, dfg = g_(mass), g_(fg)
    wo = torch.zeros(3, device=DEV)
    hip.check(lib.rmem_attn_mass_reduce(dmass.data_ptr(), N, 3, dfg.data_ptr(), wo.data_ptr(), st),
              "mass_reduce")
    assert (wo.cpu() - (mass * fg[:, None]).sum(0)).abs().max().item() < 1e-4


def test_id_assign_vs_golden(hip, deaot_model, golden_dir):
    """Bit-for-bit index work (label -> class gather) + fp32 LayerNorm, against the
    reference's own output (tests/golden/idassign_*.npz)."""
    import os
    from inputs import IDASSIGN_CASES, idassign_label
    lib, st = hip.load(), hip.stream_ptr()
    sd = deaot_model.state_dict()
    wt = sd["patch_wise_id_bank.weight"].permute(1, 2, 3, 0).contiguous().to(DEV)
    for (H, W) in IDASSIGN_CASES:
        gold = np.load(os.path.join(golden_dir, f"idassign_{H}x{W}.npz"))
        eh, ew = int(gold["eh"]), int(gold["ew"])
        lab = idassign_label(H, W)[0, 0].to(torch.uint8).to(DEV).contiguous()
        of = torch.zeros(eh * ew, 256, device=DEV)
        pl = hip.Planes.empty((eh * ew, 256), DEV)
        kb, g1, g2 = (sd["patch_wise_id_bank.bias"].to(DEV), sd["id_norm.weight"].to(DEV),
                      sd["id_norm.bias"].to(DEV))
        hip.check(lib.rmem_id_assign(lab.data_ptr(), H, W, wt.data_ptr(),
                                     kb.data_ptr(), 12, 17, 16, 8, eh, ew,
                                     256, g1.data_ptr(),
                                     g2.data_ptr(), 1e-5, pl.hi.data_ptr(),
                                     pl.lo.data_ptr(), 256, of.data_ptr(), 256, 1, st), "id_assign")
        torch.cuda.synchronize()
        assert np.abs(of.cpu().numpy() - gold["id_emb"]).max() < 2e-5
        assert np.abs(pl.float().cpu().numpy() - gold["id_emb"]).max() < 5e-5


def test_id_assign_full_size_and_label_edge_cases(hip, deaot_model):
    """ID assignment at the full 481x849 / 721x1281 frames against the oracle's restatement of
    one_hot_mask + patch_wise_id_bank + id_norm (utils/image.py:69-74, aot.py:67-74,111-114,
    deaot.py:65-69): a label map with every object id, the ignore label 255 and ids above
    MAX_OBJ (no one-hot channel); an all-background and an all-ignore map."""
    from oracle import lstt_ref as R
    lib, st = hip.load(), hip.stream_ptr()
    sd = {k: v.detach().float() for k, v in deaot_model.state_dict().items()}
    wt = sd["patch_wise_id_bank.weight"].permute(1, 2, 3, 0).contiguous().to(DEV)
    kb, g1, g2 = (sd["patch_wise_id_bank.bias"].to(DEV), sd["id_norm.weight"].to(DEV), sd["id_norm.bias"].to(DEV))
    rs = np.random.RandomState(3)
    for (H, W) in [(481, 849), (721, 1281)]:
        eh, ew = (H - 1) // 16 + 1, (W - 1) // 16 + 1
        coarse = rs.randint(0, 11, ((H + 23) // 24, (W + 23) // 24))
        lab = np.kron(coarse, np.ones((24, 24), dtype=np.int64))[:H, :W].astype(np.uint8)
        lab[H // 3:H // 3 + 40, W // 4:W // 4 + 90] = 255           # ignore region
        lab[5:30, 10:60] = 14                                        # id above MAX_OBJ: contributes nothing
        cases = [lab, np.zeros((H, W), np.uint8), np.full((H, W), 255, np.uint8)] if H == 481 else [lab]
        for m in cases:
            for ign in (True, False):     # False: reference-frame rule, label 255 contributes nothing
                ref = R.id_assign(torch.from_numpy(m.astype(np.float32))[None, None], sd, use_ignore=ign)
                d = torch.from_numpy(m).to(DEV).contiguous()
                of = torch.zeros(eh * ew, 256, device=DEV)
                pl = hip.Planes.empty((eh * ew, 256), DEV)
                hip.check(lib.rmem_id_assign(d.data_ptr(), H, W, wt.data_ptr(), kb.data_ptr(), 12, 17, 16, 8, eh, ew,
                                             256, g1.data_ptr(), g2.data_ptr(), 1e-5, pl.hi.data_ptr(),
                                             pl.lo.data_ptr(), 256, of.data_ptr(), 256, int(ign), st), "id_assign")
                torch.cuda.synchronize()
                err = (of.cpu() - ref).abs().max().item()
                assert err < 5e-5, (H, W, ign, err)


def test_groupnorm_nchw_relu(hip):
    """FPN support kernel: GroupNorm(8)+ReLU on NCHW against torch (fp64 statistics)."""
    rs = np.random.RandomState(9)
    for (c, h, w) in [(256, 31, 54), (128, 61, 107), (128, 7, 9)]:
        x = _rand(rs, 1, c, h, w, scale=2.0) + 0.7
        gn = torch.nn.GroupNorm(8, c)
        with torch.no_grad():
            gn.weight.copy_(_rand(rs, c) * 0.2 + 1)
            gn.bias.copy_(_rand(rs, c) * 0.3)
        ref = torch.relu(torch.nn.functional.group_norm(x.double(), 8, gn.weight.double(), gn.bias.double(), gn.eps))
        gn = gn.to(DEV)
        y = hip.groupnorm_nchw(x.to(DEV), gn, True)
        torch.cuda.synchronize()
        assert (y.cpu().double() - ref).abs().max().item() < 5e-6
        # with the producing convolution's bias folded in: GN(x + b[c]), x + b rounded once in fp32
        cb = _rand(rs, c) * 0.5
        xb = x + cb.view(1, -1, 1, 1)
        ref = torch.relu(torch.nn.functional.group_norm(xb.double(), 8, gn.weight.cpu().double(),
                                                        gn.bias.cpu().double(), gn.eps))
        y = hip.groupnorm_nchw(x.to(DEV), gn, True, conv_bias=cb.to(DEV))
        torch.cuda.synchronize()
        assert (y.cpu().double() - ref).abs().max().item() < 5e-6


@pytest.mark.gpu
@pytest.mark.parametrize("geom", [(128, 121, 213, 61, 107, True), (256, 61, 107, 31, 54, True),
                                  (256, 31, 54, 31, 54, True), (64, 30, 53, 15, 27, False), (8, 7, 9, 7, 9, False)])
def test_upsample_add_nchw(geom):
    """rmem_upsample_add_nchw == (y + bias) + F.interpolate(x, bilinear) (decoders/fpn.py:53-60)."""
    from rmem_amd import hip
    C, H, W, h, w, align = geom
    g = torch.Generator().manual_seed(C + H)
    y = torch.randn(1, C, H, W, generator=g).to(DEV)
    x = torch.randn(1, C, h, w, generator=g).to(DEV)
    b = torch.randn(C, generator=g).to(DEV)
    want = (y + b.view(1, -1, 1, 1)) + (x if (h, w) == (H, W) else
                                        F.interpolate(x, size=(H, W), mode="bilinear", align_corners=align))
    got = hip.upsample_add_nchw_(y.clone(), b, x, align)
    err = float((got - want).abs().max())
    print("upsample_add max abs err", err)
    assert err < 2e-6
    got2 = hip.upsample_add_nchw_(y.clone(), None, x, align)
    assert float((got2 - (want - b.view(1, -1, 1, 1))).abs().max()) < 4e-6
    y0 = y.clone()
    got3 = hip.upsample_add_nchw_(y, b, x, align, inplace=False)      # out of place: y untouched
    assert torch.equal(got3, got) and torch.equal(y, y0)


@pytest.mark.gpu
@pytest.mark.parametrize("geom", [(64, 121, 213), (256, 31, 54), (7, 5, 3), (3, 1, 5), (1024, 31, 54),
                                  (2, 64, 121, 213), (3, 7, 5, 3)])
def test_bias_act_nchw(geom):
    """rmem_bias_act_nchw == relu(x + bias[c] (+ residual)) bit for bit (one fp32 add per term, same
    order as the separate PyTorch ops); H*W not a multiple of 4 makes float4 groups straddle channels."""
    from rmem_amd import hip
    B_, C_, H, W = geom if len(geom) == 4 else (1,) + tuple(geom)        # batch > 1: several frames per encoder pass
    g = torch.Generator().manual_seed(C_ * 7 + W)
    x = torch.randn(B_, C_, H, W, generator=g).to(DEV)
    b = torch.randn(C_, generator=g).to(DEV)
    r = torch.randn(B_, C_, H, W, generator=g).to(DEV)
    for res in (None, r):
        for relu in (True, False):
            want = x + b.view(1, -1, 1, 1)
            if res is not None:
                want = want + res
            if relu:
                want = torch.relu(want)
            got = hip.bias_act_nchw_(x.clone(), b, res, relu)
            assert torch.equal(got, want), (geom, res is not None, relu)


# ------------------------------------------------------------------ fused memory read (read64.hip)
def _block16(Vf):
    """channel-major V^T [S][C][Npad] -> blocked-16 [S][Npad/16][C][16] (the layout rmem_attn_read reads)."""
    S, Cn, Np = Vf.shape
    return Vf.permute(0, 2, 1).reshape(S, Np // 16, 16, Cn).permute(0, 1, 3, 2).contiguous()


def _run_read(hip, mode, T, N, Npad, K, Vb, slot_map, Q, bias, U, h, w, R, ksplits, want_mass=True, ldr=None, rcs=0,
              uneven=None, fuse=False):
    """K: [S][Npad][128] planes, Vb: blocked-16 planes [S][Npad/16][1024][16], Q planes [Npad][128]."""
    lib, st = hip.load(), hip.stream_ptr()
    part = torch.full((ksplits, Npad, 1024), float("nan"), device=DEV)     # every valid row must be written
    ml = torch.full((ksplits, Npad, 2), float("nan"), device=DEV)
    lslot = torch.full((ksplits, Npad, T, 2), float("nan"), device=DEV) if want_mass else None
    G = torch.zeros(N, 1024, device=DEV)
    mass = torch.zeros(N, T, device=DEV)
    sm = torch.tensor(slot_map, dtype=torch.int32, device=DEV) if slot_map is not None else None
    ra = hip.ReadArgs()
    ra.mode, ra.qh, ra.ql = mode, Q.hi.data_ptr(), Q.lo.data_ptr()
    ra.kh, ra.kl, ra.k_slot_stride = K.hi.data_ptr(), K.lo.data_ptr(), Npad * 128
    ra.vh, ra.vl, ra.v_slot_stride = Vb.hi.data_ptr(), Vb.lo.data_ptr(), 1024 * Npad
    ra.slot_map = sm.data_ptr() if sm is not None else None
    ra.T, ra.N, ra.Npad, ra.ncols, ra.scale = T, N, Npad, 1024, 1.0 / math.sqrt(128)
    ra.bias = bias.data_ptr() if bias is not None else None
    if R is not None:
        ra.R, ra.ldr, ra.rcs = R.data_ptr(), (ldr if ldr is not None else R.shape[1]), rcs
    ra.h, ra.w, ra.ksplits = h, w, ksplits
    if uneven is not None:
        ra.nfull, ra.pf = uneven
    ra.part, ra.ml = part.data_ptr(), ml.data_ptr()
    ra.lslot = lslot.data_ptr() if want_mass else None
    if fuse:                   # one split: the read writes the gated aggregate itself (rmem_read_args.gate / gout), no combine
        assert ksplits == 1 and not want_mass
        G.fill_(7.0)
        ra.gate, ra.ldgate, ra.gout, ra.ldgout = U.data_ptr(), 1024, G.data_ptr(), 1024
        hip.check(lib.rmem_attn_read(C.byref(ra), st), "read")
        torch.cuda.synchronize()
        return G, mass, part, ml
    hip.check(lib.rmem_attn_read(C.byref(ra), st), "read")
    ca = hip.ReadCombineArgs()
    ca.T, ca.N, ca.Npad, ca.ncols, ca.ksplits = T, N, Npad, 1024, ksplits
    ca.part, ca.ml, ca.lslot = part.data_ptr(), ml.data_ptr(), (lslot.data_ptr() if want_mass else None)
    ca.U, ca.ldu, ca.G, ca.ldg = U.data_ptr(), 1024, G.data_ptr(), 1024
    ca.mass = mass.data_ptr() if want_mass else None
    hip.check(lib.rmem_attn_read_combine(C.byref(ca), st), "read_combine")
    torch.cuda.synchronize()
    return G, mass, part, ml


def _bank_case(rs, T, h, w, spike=False, rising=False):
    N = h * w
    Npad = (N + 127) // 128 * 128
    S = T + 2
    slot_map = [int(x) for x in rs.permutation(S)[:T]]
    Kf = torch.zeros(S, Npad, 128)
    Vf = torch.zeros(S, 1024, Npad)
    Kf[:, :N] = _rand(rs, S, N, 128, scale=1.5)
    Vf[:, :, :N] = _rand(rs, S, 1024, N)
    Kf[:, N:] = 37.0      # padding rows must be ignored (masked), not merely zero
    Vf[:, :, N:] = -53.0
    Qf = torch.zeros(Npad, 128)
    Qf[:N] = _rand(rs, N, 128, scale=1.5)
    if rising:
        # every 64-key tile scores ~14 bits above the one before: the row reference is raised and the accumulators are
        # rescaled on EVERY tile (RD_BUMP = 12), through all four flag / factor slots, across slot boundaries
        Qf[:N] += 1.0
        for ti, s_ in enumerate(slot_map):
            Kf[s_, :N] += 0.0134 * (ti * N + torch.arange(N, dtype=torch.float32))[:, None]
    if spike:
        # force the rescale path: a few keys late in the bank score far above everything a query has seen before
        # (the row maximum jumps by much more than RD_BUMP = 12 mid-split)
        for qq in (0, 5, N // 2, N - 1):
            for (tt, kk) in ((T - 1, N - 3), (T // 2, N // 2 + 1)):
                Kf[slot_map[tt], kk] = Qf[qq] * (1.0 + 0.5 * (tt + 1))
    bias = _rand(rs, N, T, scale=3.0)
    U = _rand(rs, N, 1024)
    Kl = torch.stack([Kf[s, :N] for s in slot_map]).double()            # [T][N][128]
    Vl = torch.stack([Vf[s, :, :N].t() for s in slot_map]).double()     # [T][N][1024]
    S_ = torch.einsum("qc,tkc->qtk", Qf[:N].double(), Kl) + bias.double()[:, :, None]
    S_ = S_ / math.sqrt(128)
    A = torch.softmax(S_.reshape(N, T * N), dim=1).reshape(N, T, N)
    ref = torch.einsum("qtk,tkc->qc", A, Vl) * U.double()
    return N, Npad, slot_map, Kf, Vf, Qf, bias, U, A, ref, S_


@pytest.mark.parametrize("ksplits", [1, 3, 9])
@pytest.mark.parametrize("T,h,w", [(1, 9, 13), (3, 9, 13), (5, 12, 17), (4, 31, 54), (16, 5, 7)])
def test_read_bank(hip, ksplits, T, h, w):
    """Fused long-term / self read against fp64: softmax(scale*(Q.K^T + bias)) . V * U and the per-slot
    attention mass, with slot-map permutation and poisoned padding.  (4, 31, 54) is the full
    BASELINE.json configs[1] size (N=1674, 6696 keys); (16, 5, 7) is the largest bank the ABI takes on the smallest
    golden geometry: one key tile per slot, so each slot is scored by ONE of the two wave groups and nine splits leave
    units of one and two tiles."""
    if h * w > 1000 and ksplits == 1:
        pytest.skip("one split at full size only repeats the small cases")
    rs = np.random.RandomState(T * 100 + h)
    N, Npad, slot_map, Kf, Vf, Qf, bias, U, A, ref, _ = _bank_case(rs, T, h, w)
    G, mass, part, ml = _run_read(hip, 0, T, N, Npad, _planes(hip, Kf), _planes(hip, _block16(Vf)), slot_map,
                                  _planes(hip, Qf), bias.to(DEV), U.to(DEV), h, w, None, ksplits)
    err = (G.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    merr = (mass.cpu().double() - A.sum(dim=2)).abs().max().item()
    print(f"fused bank read T={T} {h}x{w} ksplits={ksplits}: G rel err {err:.2e}, mass err {merr:.2e}")
    assert err < 5e-5 and merr < 1e-5, (err, merr)
    assert torch.isfinite(ml[:, :N]).all()
    live = ml[:, :N, 1] > 0                      # splits without a key tile leave their partial unwritten
    assert torch.isfinite(part[:, :N][live]).all()


@pytest.mark.parametrize("T,h,w,ks,nfull,pf", [(4, 31, 54, 9, 7, 14), (4, 31, 54, 8, 7, 15), (3, 9, 13, 4, 2, 2),
                                              (5, 12, 17, 6, 3, 4), (2, 9, 13, 3, 1, 1)])
def test_read_bank_uneven_splits(hip, T, h, w, ks, nfull, pf):
    """rmem_read_args.nfull / pf: the first nfull key splits hold pf tiles each, the others share the rest.  Against
    fp64 like test_read_bank (output, attention mass), every partial of a live split written; and invalid geometries
    (no short split, full pieces covering every tile, mode 1) are refused."""
    rs = np.random.RandomState(T * 100 + h + ks)
    N, Npad, slot_map, Kf, Vf, Qf, bias, U, A, ref, _ = _bank_case(rs, T, h, w)
    G, mass, part, ml = _run_read(hip, 0, T, N, Npad, _planes(hip, Kf), _planes(hip, _block16(Vf)), slot_map,
                                  _planes(hip, Qf), bias.to(DEV), U.to(DEV), h, w, None, ks, uneven=(nfull, pf))
    err = (G.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    merr = (mass.cpu().double() - A.sum(dim=2)).abs().max().item()
    print(f"uneven splits T={T} {h}x{w} ks={ks} nfull={nfull} pf={pf}: G rel err {err:.2e}, mass err {merr:.2e}")
    assert err < 5e-5 and merr < 1e-5, (err, merr)
    assert (ml[:, :N, 1] > 0).all()              # every split holds keys
    assert torch.isfinite(part[:, :N]).all()
    lib, st = hip.load(), hip.stream_ptr()
    tiles = T * ((N + 63) // 64)
    for bad in ((ks, pf), (nfull, tiles), (-1, pf)):
        with pytest.raises(hip.RmemError):
            _run_read(hip, 0, T, N, Npad, _planes(hip, Kf), _planes(hip, _block16(Vf)), slot_map, _planes(hip, Qf),
                      bias.to(DEV), U.to(DEV), h, w, None, ks, uneven=bad)


@pytest.mark.parametrize("ksplits", [1, 3])
def test_read_bank_rising_logits(hip, ksplits):
    """The online reference of the fused read under its worst case: logits that rise by ~14 bits per 64-key tile, so every
    tile raises m and rescales O, the row sums and the parked per-slot sums (read64.hip: score_p1 / pv_phase / follow)."""
    T, h, w = 4, 12, 17
    rs = np.random.RandomState(11)
    N, Npad, slot_map, Kf, Vf, Qf, bias, U, A, ref, _ = _bank_case(rs, T, h, w, rising=True)
    G, mass, part, ml = _run_read(hip, 0, T, N, Npad, _planes(hip, Kf), _planes(hip, _block16(Vf)), slot_map,
                                  _planes(hip, Qf), bias.to(DEV), U.to(DEV), h, w, None, ksplits)
    err = (G.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    merr = (mass.cpu().double() - A.sum(dim=2)).abs().max().item()
    print(f"rising logits ksplits={ksplits}: G rel err {err:.2e}, mass err {merr:.2e}")
    assert torch.isfinite(G).all() and err < 5e-5 and merr < 1e-5, (err, merr)


def test_read_bank_720p_k8_properties(hip):
    """BASELINE.json configs[2] size (720p: 46x81 = 3726 tokens, K=8: 29808 keys), where a fp64 reference
    is too slow for a test: size-independent properties of the fused long-term read.
    (a) the per-slot attention mass of every query sums to 1; (b) doubling V doubles the output (split
    planes, MFMA products and fp32 sums all scale exactly by 2; low-plane values below 2^-14 are fp16
    subnormals with resolution 2^-24, hence 2e-7 instead of bit for bit); (c) storing the bank slots in
    another physical order, with the slot map compensating, changes nothing -- bit for bit; (d) the
    number of key splits only re-associates fp32 sums: 1e-5 relative."""
    T, h, w = 8, 46, 81
    rs = np.random.RandomState(7)
    N = h * w
    Npad = (N + 127) // 128 * 128
    S = T + 2
    Kf = torch.zeros(S, Npad, 128)
    Vf = torch.zeros(S, 1024, Npad)
    Kf[:, :N] = _rand(rs, S, N, 128, scale=1.5)
    Vf[:, :, :N] = _rand(rs, S, 1024, N)
    Qf = torch.zeros(Npad, 128)
    Qf[:N] = _rand(rs, N, 128, scale=1.5)
    bias = _rand(rs, N, T, scale=3.0).to(DEV)
    U = _rand(rs, N, 1024).to(DEV)
    Qp = _planes(hip, Qf)
    map_a = [int(x) for x in rs.permutation(S)[:T]]
    run = lambda K_, V_, m_, ks: _run_read(hip, 0, T, N, Npad, _planes(hip, K_), _planes(hip, _block16(V_)), m_, Qp,
                                           bias, U, h, w, None, ks)
    G, mass, *_ = run(Kf, Vf, map_a, 4)
    assert torch.isfinite(G).all()
    assert (mass.sum(dim=1) - 1).abs().max().item() < 2e-6                      # (a)
    G2, *_ = run(Kf, 2 * Vf, map_a, 4)
    assert (G2 - 2 * G).abs().max().item() <= 2e-7 * (2 * G).abs().max().item()   # (b)
    perm = [int(x) for x in rs.permutation(S)]                                  # new physical position of slot s
    Kp, Vp = torch.zeros_like(Kf), torch.zeros_like(Vf)
    for s_old, s_new in enumerate(perm):
        Kp[s_new], Vp[s_new] = Kf[s_old], Vf[s_old]
    G3, mass3, *_ = run(Kp, Vp, [perm[s] for s in map_a], 4)
    assert torch.equal(G3, G) and torch.equal(mass3, mass)                      # (c)
    G4, mass4, *_ = run(Kf, Vf, map_a, 2)
    assert (G4 - G).abs().max().item() <= 1e-5 * G.abs().max().item()          # (d)
    assert (mass4 - mass).abs().max().item() < 2e-6


def test_read_bank_logits_and_rescale(hip):
    """(a) the statistics the kernel writes ARE the logits' log-sum-exp: m + log(l) within 1e-3 of fp64
    (north star: attention logits within 1e-3); (b) keys whose score jumps far above the running
    maximum in the middle of a split force the deferred rescale (segment flush through the partial
    buffer): outputs and mass must still match fp64."""
    rs = np.random.RandomState(77)
    for spike in (False, True):
        T, h, w = 4, 20, 23
        N, Npad, slot_map, Kf, Vf, Qf, bias, U, A, ref, S_ = _bank_case(rs, T, h, w, spike=spike)
        for ksplits in (1, 2):
            G, mass, part, ml = _run_read(hip, 0, T, N, Npad, _planes(hip, Kf), _planes(hip, _block16(Vf)), slot_map,
                                          _planes(hip, Qf), bias.to(DEV), U.to(DEV), h, w, None, ksplits)
            err = (G.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
            merr = (mass.cpu().double() - A.sum(dim=2)).abs().max().item()
            mlc = ml[:, :N].cpu().double()
            lse = torch.logsumexp(mlc[..., 0] + torch.log(mlc[..., 1]), dim=0)
            lse_ref = torch.logsumexp(S_.reshape(N, -1), dim=1)
            lerr = (lse - lse_ref).abs().max().item()
            print(f"spike={spike} ksplits={ksplits}: G rel err {err:.2e} mass {merr:.2e} logsumexp err {lerr:.2e}")
            assert err < 5e-5 and merr < 1e-5 and lerr < 1e-3, (spike, ksplits, err, merr, lerr)


def _read_logits(hip, mode, T, N, Npad, K, Q, bias, h, w, R, ksplits, ldr=None, rcs=0):
    """rmem_attn_read_trace with dbg_logits: every pre-softmax logit the fused read computed -> [N][T*N] (NaN = not
    written: keys outside the image / window)."""
    lib, st = hip.load(), hip.stream_ptr()
    Vb = hip.Planes.empty((T, Npad // 16, 1024, 16), DEV)
    part = torch.zeros(ksplits, Npad, 1024, device=DEV)
    ml = torch.zeros(ksplits, Npad, 2, device=DEV)
    out = torch.full((N, T * N), float("nan"), device=DEV)
    nq = (N + 63) // 64
    trace = torch.zeros(8 * ((nq * ksplits + 7) // 8) * 64, dtype=torch.int64, device=DEV)
    ra = hip.ReadArgs()
    ra.mode, ra.qh, ra.ql = mode, Q.hi.data_ptr(), Q.lo.data_ptr()
    ra.kh, ra.kl, ra.k_slot_stride = K.hi.data_ptr(), K.lo.data_ptr(), Npad * 128
    ra.vh, ra.vl, ra.v_slot_stride = Vb.hi.data_ptr(), Vb.lo.data_ptr(), 1024 * Npad
    ra.T, ra.N, ra.Npad, ra.ncols, ra.scale = T, N, Npad, 1024, 1.0 / math.sqrt(128)
    ra.bias = bias.data_ptr() if bias is not None else None
    if R is not None:
        ra.R, ra.ldr, ra.rcs = R.data_ptr(), (ldr if ldr is not None else R.shape[1]), rcs
    ra.h, ra.w, ra.ksplits = h, w, ksplits
    ra.part, ra.ml = part.data_ptr(), ml.data_ptr()
    ra.dbg_logits, ra.dbg_ld = out.data_ptr(), T * N
    hip.check(lib.rmem_attn_read_trace(C.byref(ra), trace.data_ptr(), st), "read_trace")
    torch.cuda.synchronize()
    return out.cpu()


LOGIT_CASES = [(0, 1, 5, 7), (1, 2, 5, 7), (1, 4, 8, 11), (2, 5, 8, 11), (1, 8, 5, 7), (1, 3, 9, 13)]


@pytest.mark.parametrize("case", LOGIT_CASES, ids=lambda c: "l%d_T%d_%dx%d" % c)
def test_read_logits_per_logit_vs_reference(hip, case, deaot_model, golden_dir):
    """"Attention logits within 1e-3" (BASELINE.json north_star) checked PER LOGIT against the reference: the block fixtures
    hold the inputs of the reference's two softmax calls -- GatedPropagation's QK = (Q / T) @ K with the temporal positional
    embedding (attention.py:184-187, transformer.py:1140-1175) and LocalGatedPropagation's qk + relative bias over the
    15 x 15 window (attention.py:334-344) -- recorded by make_golden.py:gen_blocks.  The fused read's debug entry
    (rmem_attn_read_trace, rmem_read_args.dbg_logits) dumps every logit it computes from the same Q (the reference's
    curr_K output), the same PE-free bank keys and the PE / relative-position biases formed as rmem_amd/lstt.py forms them."""
    import os
    from inputs import block_case_name, block_inputs
    from rmem_amd.lstt import temporal_pe_rows
    layer, T, h, w = case
    gold = np.load(os.path.join(golden_dir, block_case_name(layer, T, h, w, False) + ".npz"))
    i = block_inputs(layer, T, h, w, False)
    sd = {k: v.detach() for k, v in deaot_model.state_dict().items()}
    N = h * w
    Npad = (N + 127) // 128 * 128
    Q = torch.from_numpy(gold["curr_K"])                               # the block's Q projection = curr_K (transformer.py:1234)
    cur_pe, mem_pe = sd["cur_pos_emb"][0], sd["mem_pos_emb"]
    pad = lambda x: torch.cat([x, torch.zeros(Npad - N, x.shape[1])], 0)
    # ---- long-term read: Q + cur_pe against the PE-free bank, the memory PE as a per-(query, slot) bias
    Qpe = Q + cur_pe[None]
    rows = temporal_pe_rows(T)
    bias = (Qpe.double() @ mem_pe[rows].double().t()).float().to(DEV).contiguous()                      # [N][T]
    Kb = torch.stack([pad(i["bank_K"][t]) for t in range(T)])
    got = _read_logits(hip, 0, T, N, Npad, _planes(hip, Kb), _planes(hip, pad(Qpe)), bias, h, w, None, min(3, T))
    ref = torch.from_numpy(gold["lt_logits"])
    assert not torch.isnan(got).any(), "a long-term logit was not computed"
    err = (got - ref).abs().max().item()
    # ---- windowed read: unscaled Q for the relative bias (attention.py:314), scaled Q . K inside the 15 x 15 window
    p = f"LSTT.layers.{layer}.short_term_attn.relative_emb_k."
    R = (Q.double() @ sd[p + "weight"].reshape(225, 128).double().t() + sd[p + "bias"].double()).float().to(DEV).contiguous()
    gotw = _read_logits(hip, 1, 1, N, Npad, _planes(hip, pad(i["short_K"])[None]), _planes(hip, pad(Q)), None, h, w, R, 1)
    refw = torch.from_numpy(gold["st_logits"])                         # [225][N], -1e8 outside the image
    errw, n_in = 0.0, 0
    for q in range(N):
        qy, qx = divmod(q, w)
        for o in range(225):
            ky, kx = qy + o // 15 - 7, qx + o % 15 - 7
            inside = 0 <= ky < h and 0 <= kx < w
            assert (refw[o, q].item() > -1e7) == inside
            if inside:
                g = gotw[q, ky * w + kx].item()
                assert g == g, f"window logit (q={q}, o={o}) was not computed"
                errw = max(errw, abs(g - refw[o, q].item()))
                n_in += 1
    assert int((~torch.isnan(gotw)).sum()) == n_in, "logits outside the window were computed"
    print(f"l{layer} T={T} {h}x{w}: max |HIP logit - reference logit| long-term {err:.2e} (|logit| up to "
          f"{ref.abs().max().item():.1f}), windowed {errw:.2e} over {n_in} in-window logits")
    assert err < 1e-3 and errw < 1e-3, (err, errw)


def test_read_logits_per_logit_at_480p_k4_vs_reference(hip, deaot_model, golden_dir):
    """The per-logit check at a BENCHMARKED size: 31 x 54 = 1674 tokens, a bank of four slots (BASELINE.json configs[1]).
    tests/golden/block_l1_T4_31x54_rows.npz (make_golden.py:gen_block_fullsize) holds the reference block's Q projection and,
    for 48 query rows spread over the image, its own pre-softmax logits of the long-term read ([rows][4 * 1674]) and of the
    windowed read ([225][rows]); the fused read's debug entry dumps what it computes for the whole launch with the
    benchmark's seven key splits.  Every sampled logit within 1e-3 (north star); measured ~1e-5."""
    import os
    from inputs import block_inputs
    from rmem_amd.lstt import temporal_pe_rows
    layer, T, h, w = 1, 4, 31, 54
    gold = np.load(os.path.join(golden_dir, f"block_l{layer}_T{T}_{h}x{w}_rows.npz"))
    rows = torch.from_numpy(gold["rows"]).long()
    i = block_inputs(layer, T, h, w, False)
    sd = {k: v.detach() for k, v in deaot_model.state_dict().items()}
    N = h * w
    Npad = (N + 127) // 128 * 128
    Q = torch.from_numpy(gold["curr_K"])
    cur_pe, mem_pe = sd["cur_pos_emb"][0], sd["mem_pos_emb"]
    pad = lambda x: torch.cat([x, torch.zeros(Npad - N, x.shape[1])], 0)
    Qpe = Q + cur_pe[None]
    bias = (Qpe.double() @ mem_pe[temporal_pe_rows(T)].double().t()).float().to(DEV).contiguous()
    Kb = torch.stack([pad(i["bank_K"][t]) for t in range(T)])
    got = _read_logits(hip, 0, T, N, Npad, _planes(hip, Kb), _planes(hip, pad(Qpe)), bias, h, w, None, 7)
    assert not torch.isnan(got).any(), "a long-term logit was not computed"
    ref = torch.from_numpy(gold["lt_logits_rows"])
    err = (got[rows] - ref).abs().max().item()
    p = f"LSTT.layers.{layer}.short_term_attn.relative_emb_k."
    R = (Q.double() @ sd[p + "weight"].reshape(225, 128).double().t() + sd[p + "bias"].double()).float().to(DEV).contiguous()
    gotw = _read_logits(hip, 1, 1, N, Npad, _planes(hip, pad(i["short_K"])[None]), _planes(hip, pad(Q)), None, h, w, R, 2)
    refw = torch.from_numpy(gold["st_logits_rows"])                    # [225][rows], -1e8 outside the image
    errw, n_in = 0.0, 0
    for j, q in enumerate(rows.tolist()):
        qy, qx = divmod(q, w)
        inside_keys = set()
        for o in range(225):
            ky, kx = qy + o // 15 - 7, qx + o % 15 - 7
            inside = 0 <= ky < h and 0 <= kx < w
            assert (refw[o, j].item() > -1e7) == inside
            if inside:
                g = gotw[q, ky * w + kx].item()
                assert g == g, f"window logit (q={q}, o={o}) was not computed"
                errw = max(errw, abs(g - refw[o, j].item()))
                inside_keys.add(ky * w + kx)
                n_in += 1
        assert int((~torch.isnan(gotw[q])).sum()) == len(inside_keys), f"query {q}: logits outside its window were computed"
    print(f"480p K=4 (l{layer}, T={T}, {h}x{w}, {len(rows)} query rows): max |HIP logit - reference logit| long-term {err:.2e} "
          f"over {ref.numel()} logits (|logit| up to {ref.abs().max().item():.1f}), windowed {errw:.2e} over {n_in}")
    assert err < 1e-3 and errw < 1e-3, (err, errw)
    assert err < 1e-4 and errw < 1e-4, (err, errw)        # what the split-fp16 products deliver (fp32-class)


@pytest.mark.parametrize("ksplits", [1, 4])
@pytest.mark.parametrize("h,w", [(5, 7), (9, 13), (20, 23), (31, 54)])
def test_read_window(hip, ksplits, h, w):
    """Fused short-term 15x15 windowed read against the oracle's LocalGatedPropagation core."""
    _window_case(hip, ksplits, h, w, rising=False)


@pytest.mark.parametrize("ksplits", [1, 3])
def test_read_window_rising_logits(hip, ksplits):
    """The windowed read with logits that rise by ~14 bits per 64-key tile (the online reference is raised and O rescaled on
    every tile of the band, with masked keys and rows whose window has not started yet in the same tiles)."""
    _window_case(hip, ksplits, 20, 23, rising=True)


def _window_case(hip, ksplits, h, w, rising):
    from oracle import lstt_ref as R
    rs = np.random.RandomState(h * 10 + w)
    N = h * w
    Npad = (N + 127) // 128 * 128
    q, k = _rand(rs, N, 128, scale=1.5), _rand(rs, N, 128, scale=1.5)
    if rising:
        q = q + 1.0
        k = k + 0.0134 * torch.arange(N, dtype=torch.float32)[:, None]
    v, u = _rand(rs, N, 1024), _rand(rs, N, 1024)
    rel_w, rel_b = _rand(rs, 225, 128, scale=0.15), _rand(rs, 225, scale=0.1)
    idx, inside = R.local_window_index(h, w)
    rel = q.double() @ rel_w.double().t() + rel_b.double()
    kg = k.double()[idx.clamp(min=0)] * inside.unsqueeze(-1)
    qk = torch.einsum("nc,noc->no", q.double() / math.sqrt(128), kg) + rel
    qk = qk.masked_fill(~inside, -1e8)
    attn = torch.softmax(qk, dim=1)
    vg = v.double()[idx.clamp(min=0)] * inside.unsqueeze(-1)
    ref = torch.einsum("no,noc->nc", attn, vg) * u.double()
    Kf, Vf, Qf = torch.zeros(2, Npad, 128), torch.zeros(2, 1024, Npad), torch.zeros(Npad, 128)
    Kf[1, :N], Vf[1, :, :N], Qf[:N] = k, v.t(), q
    Kf[0] = 99.0
    Rm = torch.zeros(N, 232)
    Rm[:, :225] = rel.float()
    G, _, _, _ = _run_read(hip, 1, 1, N, Npad, _planes(hip, Kf), _planes(hip, _block16(Vf)), [1], _planes(hip, Qf),
                           None, u.to(DEV), h, w, Rm.to(DEV), ksplits, want_mass=False)
    err = (G.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    print(f"fused window read {h}x{w} ksplits={ksplits}: G rel err {err:.2e}")
    assert err < 5e-5, err
    # the same bias stored by anti-diagonals (element (q, o) at 225*(q+o) + o: ldr = 225, rcs = 226), the layout
    # the LSTT uses: coalesced gathers, bit-identical result
    Rd = torch.zeros(N * 225 + 225 * 226)
    qi, oi = torch.meshgrid(torch.arange(N), torch.arange(225), indexing="ij")
    Rd[(qi * 225 + oi * 226).flatten()] = rel.float().flatten()
    G2, _, _, _ = _run_read(hip, 1, 1, N, Npad, _planes(hip, Kf), _planes(hip, _block16(Vf)), [1], _planes(hip, Qf),
                            None, u.to(DEV), h, w, Rd.to(DEV), ksplits, want_mass=False, ldr=225, rcs=226)
    assert torch.equal(G2, G)


@pytest.mark.parametrize("h,w", [(5, 7), (9, 13), (20, 23), (31, 54)])
def test_read_single_split_gates_its_own_output(hip, h, w):
    """A read in ONE key split with rmem_read_args.gate / gout set writes G = U * O / l from its own epilogue -- no partial,
    no combine launch: bit for bit what rmem_attn_read + rmem_attn_read_combine give for one split (windowed read and the
    T = 1 bank read), every valid row written, nothing else touched, the partial buffer left alone."""
    rs = np.random.RandomState(h * 31 + w)
    N = h * w
    Npad = (N + 127) // 128 * 128
    Kf, Qf = torch.zeros(Npad, 128), torch.zeros(Npad, 128)
    Kf[:N], Qf[:N] = _rand(rs, N, 128, scale=1.5), _rand(rs, N, 128, scale=1.5)
    Kf[N:] = 37.0
    Vf = torch.full((1, 1024, Npad), -53.0)
    Vf[0, :, :N] = _rand(rs, 1024, N)
    u = _rand(rs, N, 1024).to(DEV)
    Rm = _rand(rs, N, 225, scale=0.3).to(DEV).contiguous()
    for mode in (1, 0):
        a = (hip, mode, 1, N, Npad, _planes(hip, Kf[None]), _planes(hip, _block16(Vf)), [0], _planes(hip, Qf), None, u, h, w,
             Rm if mode == 1 else None, 1)
        G0, _, _, ml0 = _run_read(*a, want_mass=False)
        G1, _, part1, ml1 = _run_read(*a, want_mass=False, fuse=True)
        assert torch.equal(G0.view(torch.int32), G1.view(torch.int32)), (mode, (G0 - G1).abs().max().item())
        assert torch.equal(ml0[:, :N].view(torch.int32), ml1[:, :N].view(torch.int32))
        assert bool(torch.isnan(part1).all()), "a fused read wrote a partial"


def test_linear_blocked_output(hip):
    """rmem_linear with pa_blocked: silu(X.W^T + b) written as blocked-16 planes at a column window of
    a [Npad/16][1024][16] tensor (how V | ID_V of the bank are produced), nbatch = 2 included."""
    rs = np.random.RandomState(11)
    ntok, Np = 150, 256
    X, W, b = _rand(rs, ntok, 256), _rand(rs, 512, 256, scale=0.1), _rand(rs, 512)
    dst = hip.Planes.empty((Np // 16, 1024, 16), DEV)
    hip.linear(_planes(hip, X), _planes(hip, W), ntok, 512, 256, ldx=256, ldy=256, bias=b.to(DEV), act=1,
               pa=dst, ldpa=1024, pa_blocked=True, pa_off=512 * 16, tile=64)
    torch.cuda.synchronize()
    ref = _silu(X.double() @ W.double().t() + b.double())                  # [ntok][512]
    got = dst.float().cpu().permute(0, 2, 1).reshape(Np, 1024)             # [token][col]
    assert (got[:ntok, 512:].double() - ref).abs().max().item() < 5e-5 * ref.abs().max().item()
    assert torch.all(got[:, :512] == 0) and torch.all(got[ntok:] == 0)
    # two blocks of the weight on two halves of the input (self-attention V1 | V2)
    S, W2, b2 = _rand(rs, ntok, 512), _rand(rs, 1024, 256, scale=0.1), _rand(rs, 1024)
    dst = hip.Planes.empty((Np // 16, 1024, 16), DEV)
    hip.linear(_planes(hip, S), _planes(hip, W2), ntok, 512, 256, ldx=512, ldy=256, bias=b2.to(DEV), act=1,
               pa=dst, ldpa=1024, pa_blocked=True, nbatch=2, bsx=256, bsy=512 * 256, bsbias=512, bspa=512 * 16,
               tile=64)
    torch.cuda.synchronize()
    ref = torch.cat([_silu(S[:, :256].double() @ W2[:512].double().t() + b2[:512].double()),
                     _silu(S[:, 256:].double() @ W2[512:].double().t() + b2[512:].double())], 1)
    got = dst.float().cpu().permute(0, 2, 1).reshape(Np, 1024)
    assert (got[:ntok].double() - ref).abs().max().item() < 5e-5 * ref.abs().max().item()


@pytest.mark.parametrize("nsplit", [3, 1])
def test_linear_stream_equals_tile_kernels(hip, nsplit):
    """The streaming kernel (tile 0: one persistent workgroup per CU, 64 x 128 tiles, LDS-DMA ring) against the
    tile-per-workgroup kernels (tile 64) on the launch shapes of the memory path -- bias / SiLU, two K segments,
    nbatch, split-K partials, blocked-16 planes, plane + addvec outputs, the anti-diagonal column stride, accumulate
    with two destinations, ragged M / N, one grouped launch of six problems: every output bit for bit (each element
    sees the same MFMAs in the same order), and nothing written outside the outputs."""
    rs = np.random.RandomState(17)
    P = lambda x: _planes(hip, x)

    def run(tile):
        out = {}
        # (a) grouped launch in the shape of a layer's front projections: shared X, ragged N, four epilogues
        M, K = 333, 256
        X, X2 = P(_rand(rs0, M, K)), P(_rand(rs0, M, K))
        Wq, Wr, Wpe, Wv, Wu = (P(_rand(rs0, n, K, scale=0.1)) for n in (128, 225, 5, 512, 512))
        bq, br, bpe, bv, bu, addv = (_rand(rs0, n).to(DEV) for n in (128, 225, 5, 512, 512, 128))
        Mp = (M + 15) // 16 * 16
        pa, pb = hip.Planes.empty((M, 128), DEV), hip.Planes.empty((M, 128), DEV)
        R = torch.full((M * 225 + 225 * 226 + 7,), 3.0, device=DEV)
        pe = torch.full((M, 5), 3.0, device=DEV)
        vb = hip.Planes.empty((Mp // 16, 1024, 16), DEV)
        U = torch.full((M, 1024), 3.0, device=DEV)
        hip.linear_grouped([
            hip.linear(X, Wq, M, 128, K, ldx=K, ldy=K, bias=bq, pa=pa, ldpa=128, pb=pb, ldpb=128, addvec=addv,
                       nsplit=nsplit, tile=tile, launch=False),
            hip.linear(X, Wr, M, 225, K, ldx=K, ldy=K, bias=br, d0=R.data_ptr(), ldd0=225, d0_cs=226, nsplit=nsplit,
                       tile=tile, launch=False),
            hip.linear(X, Wpe, M, 5, K, ldx=K, ldy=K, bias=bpe, d0=pe.data_ptr(), ldd0=5, nsplit=nsplit, tile=tile,
                       launch=False),
            hip.linear(X, Wv, M, 512, K, ldx=K, ldy=K, bias=bv, act=1, pa=vb, ldpa=1024, pa_blocked=True, nsplit=nsplit,
                       tile=tile, launch=False),
            hip.linear(X, Wu, M, 512, K, ldx=K, ldy=K, bias=bu, act=1, d0=U.data_ptr(), ldd0=1024, nsplit=nsplit,
                       tile=tile, launch=False),
            hip.linear(X2, Wu, M, 512, K, ldx=K, ldy=K, bias=bv, act=1, d0=U.data_ptr() + 512 * 4, ldd0=1024,
                       nsplit=nsplit, tile=tile, launch=False)])
        out.update(pa_hi=pa.hi, pa_lo=pa.lo, pb_hi=pb.hi, pb_lo=pb.lo, R=R, pe=pe, vb_hi=vb.hi, vb_lo=vb.lo, U=U)
        # (b) split-K over two K segments -> partials; (c) nbatch = 2 block-diagonal; (d) accumulate into two destinations
        M2, K2 = 150, 1024
        Ya, Yb, Wp = P(_rand(rs0, M2, 512)), P(_rand(rs0, M2, 512)), P(_rand(rs0, 200, K2, scale=0.1))
        bp = _rand(rs0, 200).to(DEV)
        parts = torch.full((4, M2, 200), 3.0, device=DEV)
        hip.linear(Ya, Wp, M2, 200, K2, ldx=512, ldy=K2, x2=Yb, ldx2=512, kx_split=512, bias=bp, nsplit=nsplit, tile=tile,
                   ksplits=4, parts=parts, part_stride=M2 * 200)
        S_, W12, b12 = P(_rand(rs0, M2, 512)), P(_rand(rs0, 2 * 192, 256, scale=0.1)), _rand(rs0, 2 * 192).to(DEV)
        D12 = torch.full((M2, 2 * 192), 3.0, device=DEV)
        hip.linear(S_, W12, M2, 192, 256, ldx=512, ldy=256, bias=b12, act=1, d0=D12.data_ptr(), ldd0=2 * 192, nbatch=2,
                   bsx=256, bsy=192 * 256, bsbias=192, bsd=192, nsplit=nsplit, tile=tile)
        t0, t1 = _rand(rs0, M2, 96).to(DEV), _rand(rs0, M2, 104).to(DEV)
        hip.linear(Ya, Wp, M2, 200, 512, ldx=512, ldy=K2, bias=bp, d0=t0.data_ptr(), ldd0=96, d1=t1.data_ptr(), ldd1=104,
                   csplit=96, accumulate=True, nsplit=nsplit, tile=tile)
        out.update(parts=parts, D12=D12, t0=t0, t1=t1)
        torch.cuda.synchronize()
        return {k: v.clone() for k, v in out.items()}

    rs0 = np.random.RandomState(17)
    a = run(64)
    rs0 = np.random.RandomState(17)
    b = run(0)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert torch.all(a["R"][-7:] == 3.0) and torch.all(b["R"][-7:] == 3.0)


@pytest.mark.parametrize("N", [1674, 100])
def test_groupnorm2_fold_equals_layernorm_fold_then_groupnorm(hip, N):
    """rmem_groupnorm2_fold (the statistics pass of the final GroupNorm1D(2 groups), transformer.py:806-808, first sums the
    split-K partials of the last self-attention projection into tgt / tgt_id, :1231-1232) against the two launches it
    replaces -- rmem_layernorm_red2 as the fold (its planes unused) + rmem_groupnorm2: folded streams and output bit for
    bit, and against an fp64 GroupNorm."""
    lib, st = hip.load(), hip.stream_ptr()
    rs = np.random.RandomState(N)
    KS = 2
    tgt, tgi = (_rand(rs, N, 256) * 2 + 0.3).to(DEV), (_rand(rs, N, 256) * 0.7).to(DEV)
    parts = _rand(rs, KS, N, 512, scale=0.5).to(DEV)
    gm, bt = (_rand(rs, 512) * 0.2 + 1).to(DEV), (_rand(rs, 512) * 0.1).to(DEV)
    ws = torch.zeros(4 * ((N + 63) // 64), dtype=torch.float64, device=DEV)
    # reference: LayerNorm launch as the fold, then the plain GroupNorm
    a, b = tgt.clone(), tgi.clone()
    dump = hip.Planes.empty((N, 512), DEV)
    hip.check(lib.rmem_layernorm_red2(a.data_ptr(), b.data_ptr(), 256, parts.data_ptr(), parts.data_ptr() + 1024, KS, N * 512, 512,
                                      gm.data_ptr(), bt.data_ptr(), gm.data_ptr(), bt.data_ptr(), N, 256, 1e-5,
                                      dump.hi.data_ptr(), dump.lo.data_ptr(), 512, dump.hi.data_ptr() + 512, dump.lo.data_ptr() + 512, 512,
                                      st), "ln_red2")
    out_ref = torch.zeros(N, 512, device=DEV)
    hip.check(lib.rmem_groupnorm2(a.data_ptr(), b.data_ptr(), N, 256, gm.data_ptr(), bt.data_ptr(), 1e-5, ws.data_ptr(),
                                  out_ref.data_ptr(), 512, st), "gn2")
    c, d = tgt.clone(), tgi.clone()
    out = torch.zeros(N, 512, device=DEV)
    hip.check(lib.rmem_groupnorm2_fold(c.data_ptr(), d.data_ptr(), parts.data_ptr(), KS, N * 512, 512, N, 256, gm.data_ptr(),
                                       bt.data_ptr(), 1e-5, ws.data_ptr(), out.data_ptr(), 512, st), "gn2_fold")
    torch.cuda.synchronize()
    assert torch.equal(c, a) and torch.equal(d, b), "folded streams"
    assert torch.equal(out, out_ref)
    x = torch.cat([tgt.double().cpu() + parts[:, :, :256].double().sum(0).cpu(), tgi.double().cpu() + parts[:, :, 256:].double().sum(0).cpu()], 1)
    g = x.view(N, 2, 256)
    mean, var = g.mean(dim=(0, 2), keepdim=True), g.var(dim=(0, 2), unbiased=False, keepdim=True)
    ref = ((g - mean) / torch.sqrt(var + 1e-5)).view(N, 512) * gm.double().cpu() + bt.double().cpu()
    assert (out.cpu().double() - ref).abs().max().item() < 5e-6


@pytest.mark.parametrize("h,w", [(31, 54), (9, 13), (5, 7), (46, 81)])
def test_dwconv_rows_per_thread_variants_bit_identical(hip, h, w, monkeypatch):
    """The depth-wise 5 x 5 kernel with RY output rows per thread (rmem_configure("dw_rows", 2 / 3 / 4): the input window is loaded once
    for RY rows, an input element is fetched 2.9-4.3 times instead of 7.2) against the one-row kernel: both maps of the
    paired launch and the single-map launch, hi and lo planes bit for bit (the taps of an output are accumulated in the
    same order), nothing written beyond the N rows."""
    lib, st = hip.load(), hip.stream_ptr()
    rs = np.random.RandomState(h * 100 + w)
    N, C = h * w, 1024
    g0, g1 = _rand(rs, N, C).to(DEV), _rand(rs, N, C).to(DEV)
    w0, w1 = _rand(rs, 25, C, scale=0.2).to(DEV), _rand(rs, 25, C, scale=0.2).to(DEV)

    def run():
        o = [torch.full((N + 3, C), 7, dtype=torch.int16, device=DEV) for _ in range(6)]
        hip.check(lib.rmem_dwconv5x5_split2(g0.data_ptr(), g1.data_ptr(), C, w0.data_ptr(), w1.data_ptr(), h, w, C,
                                            o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), C, st), "dw2")
        hip.check(lib.rmem_dwconv5x5_split(g1.data_ptr(), C, w0.data_ptr(), h, w, C, o[4].data_ptr(), o[5].data_ptr(), C, st), "dw")
        torch.cuda.synchronize()
        return o

    try:
        hip.configure("dw_rows", 0)
        ref = run()
        for ry in (2, 3, 4):
            hip.configure("dw_rows", ry)
            got = run()
            for i, (a, b) in enumerate(zip(ref, got)):
                assert torch.equal(a, b), (ry, i)
            assert all(bool((t[N:] == 7).all()) for t in got)
    finally:
        hip.configure("dw_rows", 2)


def test_linear_single_stage_items_many_per_workgroup(hip):
    """Shapes whose work items hold ONE k-tile stage (K = 64; split-K down to one k-tile per split) with more than four
    items per workgroup of the streaming kernel: its four-entry item ring would be overwritten before the epilogue reads
    it, so the library keeps such launches on the tile kernels (linear_stream.h: use_stream) -- tile 0 must equal tile 64
    and the fp64 product."""
    rs = np.random.RandomState(5)
    M, N = 64 * 40, 128 * 40                    # 1600 items of the 64 x 128 tiling: > 4 per CU on 256 CUs
    X, Y = _rand(rs, M, 64), _rand(rs, N, 64, scale=0.1)
    xp, yp = _planes(hip, X), _planes(hip, Y)
    outs = []
    for tile in (0, 64):
        D = torch.zeros(M, N, device=DEV)
        hip.linear(xp, yp, M, N, 64, ldx=64, ldy=64, d0=D.data_ptr(), ldd0=N, nsplit=3, tile=tile)
        outs.append(D)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    ref = (xp.float().double() @ yp.float().double().t()).float()
    assert (outs[0] - ref).abs().max().item() < 3e-5 * ref.abs().max().item()
    # split-K with one k-tile per split
    X2, Y2 = _rand(rs, M, 256), _rand(rs, 512, 256, scale=0.1)
    x2, y2 = _planes(hip, X2), _planes(hip, Y2)
    parts = []
    for tile in (0, 64):
        pt = torch.zeros(4, M, 512, device=DEV)
        hip.linear(x2, y2, M, 512, 256, ldx=256, ldy=256, nsplit=3, tile=tile, ksplits=4, parts=pt, part_stride=M * 512)
        parts.append(pt)
    torch.cuda.synchronize()
    assert torch.equal(parts[0], parts[1])
    ref2 = (x2.float().double() @ y2.float().double().t()).float()
    assert (parts[0].sum(0) - ref2).abs().max().item() < 3e-5 * ref2.abs().max().item()


@pytest.mark.parametrize("N", [35, 1674])
def test_layernorm_from_channel_major_features(hip, N):
    """rmem_layernorm_cn (norm1 of layer 0 straight from the encoder's [256][N] feature map: transpose + residual stream
    + zeroed ID stream + planes in one launch) against the transposing copy + rmem_layernorm_red it replaces: every
    output bit for bit."""
    lib, st = hip.load(), hip.stream_ptr()
    rs = np.random.RandomState(N)
    src = (_rand(rs, 256, N, scale=2.0) + 0.3).to(DEV).contiguous()
    gm, bt = (_rand(rs, 256) * 0.2 + 1).to(DEV), (_rand(rs, 256) * 0.1).to(DEV)
    x_ref = src.t().contiguous()
    pl_ref = hip.Planes.empty((N, 256), DEV)
    hip.check(lib.rmem_layernorm_red(x_ref.data_ptr(), 256, None, 0, 0, 0, gm.data_ptr(), bt.data_ptr(), N, 256, 1e-5,
                                     pl_ref.hi.data_ptr(), pl_ref.lo.data_ptr(), 256, None, 0, st), "ln_red")
    x = torch.full((N + 1, 256), 7.0, device=DEV)
    z = torch.full((N + 1, 256), 7.0, device=DEV)
    pl = hip.Planes.empty((N + 1, 256), DEV)
    hip.check(lib.rmem_layernorm_cn(src.data_ptr(), N, x.data_ptr(), z.data_ptr(), gm.data_ptr(), bt.data_ptr(), N, 256,
                                    1e-5, pl.hi.data_ptr(), pl.lo.data_ptr(), 256, st), "ln_cn")
    torch.cuda.synchronize()
    assert torch.equal(x[:N], x_ref) and torch.all(x[N] == 7.0)
    assert torch.all(z[:N] == 0) and torch.all(z[N] == 7.0)
    assert torch.equal(pl.hi[:N], pl_ref.hi) and torch.equal(pl.lo[:N], pl_ref.lo)
    assert torch.all(pl.hi[N] == 0) and torch.all(pl.lo[N] == 0)


def test_fg_weights_vs_torch(hip):
    """rmem_fg_weights = 1 - softmax(bilinear align_corners resize of the decoder logits to the token grid)[0]
    (engines/aot_engine.py:350-356) against the torch ops of the reference, on the CPU and on the GPU."""
    import torch.nn.functional as F
    lib, st = hip.load(), hip.stream_ptr()
    rs = np.random.RandomState(5)
    for (Hl, Wl, h, w) in ((121, 213, 31, 54), (181, 321, 46, 81), (25, 33, 7, 9), (5, 9, 5, 9)):
        lg = _rand(rs, 1, 11, Hl, Wl, scale=4.0)
        lg[:, 4:] = -1e10                                   # unused ids are masked like this (aot_engine.py:451-453)
        d = lg.to(DEV)
        fg = torch.zeros(h * w, device=DEV)
        hip.check(lib.rmem_fg_weights(d.data_ptr(), 11, Hl, Wl, h, w, fg.data_ptr(), st), "fg")
        ref = (1 - torch.softmax(F.interpolate(lg, size=(h, w), mode="bilinear", align_corners=True), dim=1)[:, 0]).reshape(-1)
        refg = (1 - torch.softmax(F.interpolate(d, size=(h, w), mode="bilinear", align_corners=True), dim=1)[:, 0]).reshape(-1)
        torch.cuda.synchronize()
        assert (fg.cpu() - ref).abs().max().item() < 2e-6, (Hl, Wl)
        assert (fg - refg).abs().max().item() < 2e-6, (Hl, Wl)


@pytest.mark.parametrize("cap,former", [(4, 1), (8, 1), (2, 1)])
def test_bank_policy_on_device_vs_host_rule(hip, cap, former):
    """rmem_bank_reset / rmem_bank_append / rmem_bank_policy_step against the host rule
    rmem_amd.lstt.rmem_policy_step (itself pinned to the reference's EMA / visit dictionaries by the golden
    clips): 60 long-memory updates with random attention masses -- every dropped position, the slot map, the
    frame indexes and the stored EMA / visit values must be IDENTICAL (same fp32 operations in the same order)."""
    import ctypes as C
    from rmem_amd.lstt import rmem_policy_step
    lib, st = hip.load(), hip.stream_ptr()
    rs = np.random.RandomState(cap)
    maps = torch.zeros(32, dtype=torch.int32, device=DEV)
    state = torch.zeros(C.sizeof(hip.BankState) // 4, dtype=torch.int32, device=DEV)
    res = torch.zeros(4, dtype=torch.int32, device=DEV)
    S = cap + 3
    bank, idx, ema, visits = [0], [0], {}, {}
    hip.check(lib.rmem_bank_reset(maps.data_ptr(), state.data_ptr(), 0, 0, st), "reset")
    for step in range(1, 61):
        slot = next(s for s in range(S) if s not in bank)
        hip.check(lib.rmem_bank_append(maps.data_ptr(), state.data_ptr(), slot, 5 * step, st), "append")
        bank.append(slot)
        idx.append(5 * step)
        w = (rs.rand(len(bank) - 1).astype(np.float32) ** 3) * 1674.0 + 1e-3
        if step % 7 == 0:
            w[1:] = w[1]                                     # ties: the first minimum wins
        wd = torch.from_numpy(w).to(DEV)
        hip.check(lib.rmem_bank_policy_step(maps.data_ptr(), state.data_ptr(), wd.data_ptr(), len(w), cap, former,
                                            res.data_ptr(), st), "policy")
        wn = w / w.sum(dtype=np.float32)
        drop, ema, visits = rmem_policy_step(wn, idx, ema, visits, former)
        dropped = -1
        if len(bank) > cap:
            dropped = drop
            del bank[drop]
            idx.remove(idx[drop])
        torch.cuda.synchronize()
        r = res.cpu().tolist()
        stc = hip.BankState.from_buffer_copy(state.cpu().numpy().tobytes())
        assert r[0] == step and r[1] == dropped and r[2] == len(bank) == stc.T, (step, r, dropped, bank)
        assert maps.cpu().tolist()[:len(bank)] == bank and list(stc.index)[:len(bank)] == idx, step
        for p, fi in enumerate(idx):                         # dictionaries are keyed by frame index on the host
            assert (fi in ema) == bool(stc.has_ema[p]) and stc.visits[p] == visits.get(fi, 0), (step, p)
            if fi in ema:
                assert np.float32(stc.ema[p]) == np.float32(ema[fi]), (step, p, stc.ema[p], ema[fi])
