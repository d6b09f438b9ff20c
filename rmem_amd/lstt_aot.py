"""AOT LSTT (LongShortTermTransformer, 8 heads x 32) executor on the HIP kernels.

Host-side mirror of ``LongShortTermTransformer`` / ``SimplifiedTransformerBlock``
(/root/reference/aot_plus/networks/layers/transformer.py:133-692, stage pre_vost:
MODEL_LINEAR_Q=False -> ``norm4`` short-term variant).  Same design as rmem_amd.lstt:
ring of ``cap + 2`` physical bank slots per layer (K planes [slot][Npad][256], V^T planes
[slot][256][Npad]), PE-free keys with the temporal PE as a per-(query, head, slot) logit
bias, no D2H inside a frame.  All FLOPs run in rmem_amd/csrc (mha.hip, linear.hip,
pointwise.hip); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from . import hip
from .hip import Planes
from .lstt import rmem_policy_step, temporal_pe_rows


def sine_pos_emb(h: int, w: int, device, num_pos_feats: int = 128, temperature: float = 10000.0):
    """PositionEmbeddingSine(normalize=True) (networks/layers/position.py:35-77), token-major
    [N, 256]; computed once per clip (engines/aot_engine.py:289-292)."""
    scale, eps = 2 * math.pi, 1e-6
    ys = torch.arange(h, dtype=torch.float32, device=device).view(h, 1).expand(h, w)
    xs = torch.arange(w, dtype=torch.float32, device=device).view(1, w).expand(h, w)
    y_embed = ys / (ys[-1:, :] + eps) * scale
    x_embed = xs / (xs[:, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32, device=device)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="trunc") / num_pos_feats)
    px, py = x_embed[:, :, None] / dim_t, y_embed[:, :, None] / dim_t
    px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
    py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((py, px), dim=2).reshape(h * w, 2 * num_pos_feats).contiguous()


class _W:
    pass


class AOTLSTT:
    D = 256
    HEADS = 8
    FF = 1024

    def __init__(self, model, h: int, w: int, device, nsplit: int = 3):
        hip.load()
        self.cfg = model.cfg
        self.h, self.w = int(h), int(w)
        self.N = self.h * self.w
        self.Npad = (self.N + 127) // 128 * 128
        self.dev = torch.device(device)
        self.L = self.cfg.MODEL_LSTT_NUM
        self.cap = self.cfg.FORMER_MEM_LEN + self.cfg.LATTER_MEM_LEN
        self.Tmax = self.cap + 1
        if self.Tmax > 16:
            raise hip.RmemError("bank of more than 15 slots is not supported")
        self.S = self.cap + 2
        self.nsplit = int(nsplit)
        self.scale = 1.0 / math.sqrt(self.D // self.HEADS)
        self._timing, self._events = False, []
        self._pack(model)
        self._alloc()
        self.clear_memory()

    # ------------------------------------------------------------------ weights
    def _pl(self, t):
        return Planes.from_f32(t.detach().to(self.dev, torch.float32).contiguous())

    def _f(self, t):
        return t.detach().to(self.dev, torch.float32).contiguous()

    def _pack(self, model):
        sd = model.state_dict()
        self.cur_pe = self._f(sd["cur_pos_emb"][0])
        self.mem_pe = self._f(sd["mem_pos_emb"])
        kb = sd["patch_wise_id_bank.weight"]
        self.id_ksize, self.id_ncls = kb.shape[-1], kb.shape[1]
        self.id_stride, self.id_pad = (16, 8) if self.cfg.MODEL_ALIGN_CORNERS else (16, 0)
        self.id_wt = self._f(kb.permute(1, 2, 3, 0))
        self.id_bias = self._f(sd["patch_wise_id_bank.bias"])
        self.lw = []
        for l in range(self.L):
            p = f"LSTT.layers.{l}."
            g = lambda k: sd[p + k]
            W = _W()
            for n in ("norm1", "norm2", "norm3", "norm4"):
                setattr(W, n, (self._f(g(n + ".weight")), self._f(g(n + ".bias"))))
            W.dnorm = (self._f(sd[f"LSTT.decoder_norms.{l}.weight"]), self._f(sd[f"LSTT.decoder_norms.{l}.bias"]))
            W.Wqk_s = self._pl(torch.cat([g("self_attn.linear_Q.weight"), g("self_attn.linear_K.weight")], 0))
            W.bqk_s = self._f(torch.cat([g("self_attn.linear_Q.bias"), g("self_attn.linear_K.bias")], 0))
            W.Wv_s, W.bv_s = self._pl(g("self_attn.linear_V.weight")), self._f(g("self_attn.linear_V.bias"))
            W.Wp_s, W.bp_s = self._pl(g("self_attn.projection.weight")), self._f(g("self_attn.projection.bias"))
            W.Wq, W.bq = self._pl(g("linear_Q.weight")), self._f(g("linear_Q.bias"))
            W.Wv, W.bv = self._pl(g("linear_V.weight")), self._f(g("linear_V.bias"))
            # The per-head temporal-PE bias of the long-term read, bias[q][h][t] = (Q[q] + cur_pe)_h . mem_pe[row(t)]_h
            # (transformer.py:600-631), is linear in the layer input y = norm2(tgt): with Wq folded in once in fp64 it is a
            # projection of y onto heads x T columns -- a member of Q's launch instead of a dependent launch after it.
            wq64, bq64 = g("linear_Q.weight").double(), g("linear_Q.bias").double()
            mem64, cur64 = sd["mem_pos_emb"].double(), sd["cur_pos_emb"][0].double()      # [4][256], [256]
            hd = self.D // self.HEADS
            W.pe_h = {}                                                                    # T -> (planes [heads*T][256], bias [heads*T])
            for T in range(1, self.cap + 2):
                r = temporal_pe_rows(T)
                m = mem64[r].view(T, self.HEADS, hd).permute(1, 0, 2)                      # [heads][T][hd]
                wq_h = wq64.view(self.HEADS, hd, -1)                                       # [heads][hd][256]
                w_ht = torch.einsum("htc,hck->htk", m, wq_h).reshape(self.HEADS * T, -1)
                b_ht = torch.einsum("htc,hc->ht", m, (bq64 + cur64).view(self.HEADS, hd)).reshape(self.HEADS * T)
                W.pe_h[T] = (self._pl(w_ht.float()), self._f(b_ht.float()))
            W.Wqm, W.bqm = self._pl(g("linear_QMem.weight")), self._f(g("linear_QMem.bias"))
            W.Wvm, W.bvm = self._pl(g("linear_VMem.weight")), self._f(g("linear_VMem.bias"))
            W.Wp_lt, W.bp_lt = (self._pl(g("long_term_attn.projection.weight")),
                                self._f(g("long_term_attn.projection.bias")))
            W.Wp_st, W.bp_st = (self._pl(g("short_term_attn.projection.weight")),
                                self._f(g("short_term_attn.projection.bias")))
            W.W1, W.b1 = self._pl(g("linear1.weight")), self._f(g("linear1.bias"))
            W.W2, W.b2 = self._pl(g("linear2.weight")), self._f(g("linear2.bias"))
            W.gn = (self._f(g("activation.gn.weight")), self._f(g("activation.gn.bias")))
            W.dw = self._f(g("activation.conv.weight").reshape(-1, 25).t())
            self.lw.append(W)

    # ------------------------------------------------------------------ buffers
    def _alloc(self):
        N, Np, dev, L = self.N, self.Npad, self.dev, self.L
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)
        self.pos = sine_pos_emb(self.h, self.w, dev)
        self.tgt = z(N, 256)
        self.x_pl, self.xq_pl = Planes.empty((Np, 256), dev), Planes.empty((Np, 256), dev)
        self.sQK = Planes.empty((Np, 512), dev)
        self.sVt = Planes.empty((1, 256, Np), dev)
        # Key splits, chosen for the frame (round 6; profiles/r06o_aot_ks_sweep.txt, same box, alternating, R50-AOTL 480p
        # K=4 frames/s for (long, one-slot) splits): (12, 12) 455.7 / 456.6 -- the round-4 choice, tuned on the kernel's own
        # length: 78.8 us at 8 splits, 64.5 at 12 --; (8, 8) 472; (12, 4) 476-482; (6, 4) 482.1 / 481.1; (5, 4) 482.8 / 484.0;
        # (4, 2) 481.5 / 482.7; (3, 2) 477 / 473.  The frame follows the CU-time and the split-partial bytes a launch costs, not
        # its length (the encoder stream uses what it leaves free): ~670 blocks for the bank read (6 splits at 480p: the
        # kernel itself 63.9 us against 62.7), ~450 for the one-slot reads.
        blocks = (Np // 128) * self.HEADS
        self.ks = max(1, min(16, int(round(672.0 / blocks))))
        # key splits of the one-slot reads (short-term, self); RMEM_AOT_KS="long,short" overrides both (tuning)
        self.ks_short = max(1, min(self.ks, int(round(448.0 / blocks))))
        if os.environ.get("RMEM_AOT_KS"):
            v = [int(x) for x in os.environ["RMEM_AOT_KS"].split(",")]
            self.ks = max(1, min(16, v[0]))
            self.ks_short = max(1, min(self.ks, v[1] if len(v) > 1 else v[0]))
        self.opart = z(self.ks, Np, 256)
        self.ml = z(self.ks, Np, self.HEADS, 2)
        self.opart2 = z(self.ks_short, Np, 256)               # second workspace: the short-term read beside the long-term one
        self.ml2 = z(self.ks_short, Np, self.HEADS, 2)
        self.slot_ml = z(self.ks, Np, self.HEADS, self.Tmax, 2)
        self.ao_pl = Planes.empty((Np, 256), dev)
        self.ao2_pl = Planes.empty((Np, 256), dev)            # the short-term read's output (its projection shares a launch with the long-term one's)
        # launches shared by independent members of a layer (RMEM_AOT_GROUP=0: one launch each, the round-5 schedule; bit-identical)
        self.group_launches = os.environ.get("RMEM_AOT_GROUP", "1") != "0"
        self.y_f32 = [z(N, 256) for _ in range(L)]
        self.y_pl, self.yid_pl = Planes.empty((Np, 256), dev), Planes.empty((Np, 256), dev)
        # per layer: the update's sums (curr_V + id_emb, local_V + id_emb) of all layers are formed in one launch
        self.yid_l = [Planes.empty((Np, 256), dev) for _ in range(L)]
        self.t3id_l = [Planes.empty((Np, 256), dev) for _ in range(L)]
        self.Qc = z(N, 256)
        self.Qpe = Planes.empty((Np, 256), dev)
        self.bankK = [Planes.empty((self.S, Np, 256), dev) for _ in range(L)]
        self.bankV = [Planes.empty((self.S, 256, Np), dev) for _ in range(L)]
        self.sK = [z(N, 256) for _ in range(L)]
        self.sV = [z(N, 256) for _ in range(L)]
        self.nsK = [z(N, 256) for _ in range(L)]
        self.nsV = [z(N, 256) for _ in range(L)]
        self.refV = z(N, 256)
        self.tgt3 = [z(N, 256) for _ in range(L)]
        self.t3_pl, self.t3id_pl = Planes.empty((Np, 256), dev), Planes.empty((Np, 256), dev)
        self.Ks_pl = Planes.empty((1, Np, 256), dev)
        self.Vs_pl = Planes.empty((Np, 256), dev)
        self.VsT = Planes.empty((1, 256, Np), dev)
        self.bias_h = z(N, self.HEADS, self.Tmax)
        self.z_pl = Planes.empty((Np, 256), dev)
        self.a, self.g = z(N, self.FF), z(N, self.FF)
        self.gdw_pl = Planes.empty((Np, self.FF), dev)
        self.gn_ws = z(2 * 16 * 32, dt=torch.float64)
        self.outs = [z(N, 256) for _ in range(L)]
        self.idemb = z(N, 256)
        self.mass = z(N, self.Tmax)
        self.w_out = z(self.Tmax)
        self.maps = z(32, dt=torch.int32)

    def enable_kernel_timing(self, on: bool):
        self._timing = bool(on)
        if on:
            self._events = []

    def roofline_report(self, mfma_peak_tflops: float):
        if not self._events:
            return None
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b, _ in self._events]
        T = self._events[0][2]
        flops = 2.0 * self.N * (T * self.N) * (32 + 32) * self.HEADS
        mean_ms = sum(ms) / len(ms)
        ach = flops / (mean_ms * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": f"mha_flash_kernel<{self.nsplit}> (long-term, T={T})", "achieved": ach,
                "peak": mfma_peak_tflops, "unit": "TFLOP/s", "frac": ach / mfma_peak_tflops, "traffic": None,
                "launches": len(ms), "mean_us": 1e3 * mean_ms, "algorithmic_flops_per_launch": flops}

    def clear_memory(self):
        self.bank: List[int] = []
        self._flip = 0
        self.cur = 0
        self.mass_T = 0
        self.ema: Dict[int, float] = {}
        self.visits: Dict[int, int] = {}

    # ------------------------------------------------------------------ helpers
    def _free_slot(self) -> int:
        used = set(self.bank)
        for s in range(self.S):
            if s not in used:
                return s
        raise hip.RmemError("no free bank slot")

    def _ln(self, x, gb, out: Optional[Planes], x2=None, post=None, of32=None):
        rc = hip.load().rmem_layernorm_ex(
            x.data_ptr(), 256, hip.ptr(x2), 256, gb[0].data_ptr(), gb[1].data_ptr(), self.N, 256, 1e-5,
            hip.ptr(post), 256, out.hi.data_ptr() if out else None, out.lo.data_ptr() if out else None, 256,
            hip.ptr(of32), 256, hip.stream_ptr())
        hip.check(rc, "rmem_layernorm_ex")

    def _ln_multi(self, problems):
        """Several _ln() problems -- (x, gb, out, x2, post, of32[, sum_out]) each -- in ONE launch (rmem_layernorm_multi: per problem
        the arithmetic of rmem_layernorm_ex, bit for bit)."""
        arr = (hip.LnArgs * len(problems))()
        for a, pr in zip(arr, problems):
            x, gb, out, x2, post, of32 = pr[:6]
            a.x, a.ldx, a.x2, a.ldx2 = x.data_ptr(), 256, hip.ptr(x2), 256
            a.sum_out, a.ldsum = (hip.ptr(pr[6]) if len(pr) > 6 else None), 256       # (x + x2 written back: a residual add)
            a.gamma, a.beta, a.post, a.ldpost = gb[0].data_ptr(), gb[1].data_ptr(), hip.ptr(post), 256
            a.oh, a.ol, a.ldo = (out.hi.data_ptr() if out else None), (out.lo.data_ptr() if out else None), 256
            a.of32, a.ldof = hip.ptr(of32), 256
        hip.check(hip.load().rmem_layernorm_multi(arr, len(problems), self.N, 256, 1e-5, hip.stream_ptr()),
                  "rmem_layernorm_multi")

    def _add_split(self, a, b, dst=None, out: Optional[Planes] = None):
        rc = hip.load().rmem_add_split(a.data_ptr(), hip.ptr(b), self.N * 256, hip.ptr(dst),
                                       out.hi.data_ptr() if out else None, out.lo.data_ptr() if out else None,
                                       hip.stream_ptr())
        hip.check(rc, "rmem_add_split")

    def _mha_args(self, q: Planes, ldq, q_off, k: Planes, k_off, ldk, k_slot_stride, v: Planes, slot_map_ptr, T, bias,
                  want_mass: bool, out: Optional[Planes] = None, ws: int = 0):
        """Argument blocks (flash MHA, combine) of one read -> `out` (default self.ao_pl; planes [Npad][256]) through
        workspace `ws` (0: opart / ml, 1: the second set, for a read that shares its launches with another)."""
        out = out if out is not None else self.ao_pl
        opart, ml = (self.opart, self.ml) if ws == 0 else (self.opart2, self.ml2)
        a = hip.MHAArgs()
        a.qh, a.ql, a.ldq = q.hi.data_ptr() + q_off * 2, q.lo.data_ptr() + q_off * 2, ldq
        a.kh, a.kl, a.k_slot_stride, a.ldk = k.hi.data_ptr() + k_off * 2, k.lo.data_ptr() + k_off * 2, k_slot_stride, ldk
        a.vh, a.vl, a.v_slot_stride, a.ldv = v.hi.data_ptr(), v.lo.data_ptr(), 256 * self.Npad, self.Npad
        a.slot_map, a.T, a.N, a.Npad, a.heads = slot_map_ptr, T, self.N, self.Npad, self.HEADS
        ks = self.ks if T > 1 else self.ks_short
        a.scale, a.bias, a.ksplits = self.scale, hip.ptr(bias), ks
        a.opart, a.ml = opart.data_ptr(), ml.data_ptr()
        a.slot_ml = self.slot_ml.data_ptr() if want_mass else None
        a.nsplit = self.nsplit
        c = hip.MHACombineArgs()
        c.N, c.Npad, c.heads, c.T, c.ksplits = self.N, self.Npad, self.HEADS, T, ks
        c.opart, c.ml, c.slot_ml = opart.data_ptr(), ml.data_ptr(), a.slot_ml
        c.oh, c.ol, c.of32, c.ldo = out.hi.data_ptr(), out.lo.data_ptr(), None, 256
        c.mass = self.mass.data_ptr() if want_mass else None
        return a, c

    def _mha(self, *args, want_mass: bool = False, timed: bool = False, out: Optional[Planes] = None):
        """flash MHA + combine of one read (see _mha_args)."""
        lib, st = hip.load(), hip.stream_ptr()
        a, c = self._mha_args(*args, want_mass, out=out)
        if want_mass:
            self.slot_ml.zero_()
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        hip.check(lib.rmem_mha_flash(C.byref(a), st), "rmem_mha_flash")
        if timed:
            e1.record()
            self._events.append((e0, e1, a.T))
        hip.check(lib.rmem_mha_combine(C.byref(c), st), "rmem_mha_combine")

    def _mha_pair(self, A, B, want_mass: bool):
        """Two independent reads (argument blocks of _mha_args, workspaces 0 and 1): one flash launch, one combine launch."""
        lib, st = hip.load(), hip.stream_ptr()
        if want_mass:
            self.slot_ml.zero_()
        hip.check(lib.rmem_mha_flash2(C.byref(A[0]), C.byref(B[0]), st), "rmem_mha_flash2")
        hip.check(lib.rmem_mha_combine2(C.byref(A[1]), C.byref(B[1]), st), "rmem_mha_combine2")

    def assign_identity(self, label_u8: torch.Tensor, ignore: bool = True):
        """label [H][W] uint8 -> id_emb fp32 [N][256]; AOT has no id LayerNorm (aot.py:111-114).
        ignore: see DeAOTLSTT.assign_identity (False for reference frames)."""
        H, Wd = label_u8.shape
        rc = hip.load().rmem_id_assign(
            label_u8.data_ptr(), H, Wd, self.id_wt.data_ptr(), self.id_bias.data_ptr(), self.id_ncls,
            self.id_ksize, self.id_stride, self.id_pad, self.h, self.w, 256, None, None, 1e-5, None, None, 256,
            self.idemb.data_ptr(), 256, int(bool(ignore)), hip.stream_ptr())
        hip.check(rc, "rmem_id_assign")

    # ------------------------------------------------------------------ forward
    def forward(self, emb_nc: torch.Tensor, ref_frame: bool = False):
        """LongShortTermTransformer.forward (transformer.py:199-267): returns the three
        per-layer outputs after their LayerNorms, each [N,256] fp32."""
        self.tgt.copy_(emb_nc)
        self._prepare(ref_frame)
        self._forward_device(ref_frame)
        return self._finish(ref_frame)

    def _prepare(self, ref_frame: bool):
        self.cur = self._free_slot()
        bank_map = [self.cur] if ref_frame else self.bank
        self._T = len(bank_map)
        hip.set_ints(self.maps, list(bank_map))

    def _finish(self, ref_frame: bool):
        self.mass_T = self._T
        if ref_frame:                                                                   # init_memory (:438-453)
            self.bank = [self.cur]
            self._swap_short()
            self.ema, self.visits = {}, {}
        return self.outs

    def _swap_short(self):
        self._flip ^= 1

    def graph_key(self):
        # the short-term buffers alternate between two sets, so their parity is part of the key
        return (self._T, self.cur, self._flip)

    def graph_variants(self):
        return [{"cur": c, "_flip": f} for c in range(self.S) for f in (0, 1)]

    @property
    def out(self):
        return self.outs

    def _forward_device(self, ref_frame: bool = False):
        """Capturable device part (reads self.tgt, writes self.outs)."""
        N, Np, ns = self.N, self.Npad, self.nsplit
        lib = hip.load()
        cur, T = self.cur, self._T
        map_bank = self.maps.data_ptr()
        kss = Np * 256
        sK, sV = (self.sK, self.sV) if self._flip == 0 else (self.nsK, self.nsV)        # current short memory
        nK, nV = (self.nsK, self.nsV) if self._flip == 0 else (self.sK, self.sV)        # written this frame
        for l in range(self.L):
            W = self.lw[l]
            curK, curV = self.bankK[l][cur], self.bankV[l][cur]
            grp = self.group_launches
            # -- self attention with sine PE on q, k (transformer.py:558-566).  q = k = norm1(tgt) + pos and v = norm1(tgt):
            #    one launch for the two norms, one for the two projections (independent problems of one stage)
            if grp:
                if l == 0:              # (layers 1..: issued with the previous layer's output norm, same input)
                    self._ln_multi([(self.tgt, W.norm1, self.x_pl, None, None, None),
                                    (self.tgt, W.norm1, self.xq_pl, None, self.pos, None)])
            else:
                self._ln(self.tgt, W.norm1, self.x_pl)
                self._ln(self.tgt, W.norm1, self.xq_pl, post=self.pos)
            qk = hip.linear(self.xq_pl, W.Wqk_s, N, 512, 256, ldx=256, ldy=256, bias=W.bqk_s, pa=self.sQK, ldpa=512,
                            nsplit=ns, launch=not grp)
            vs = hip.linear(W.Wv_s, self.x_pl, 256, N, 256, ldx=256, ldy=256, bias=W.bv_s, bias_per_row=True,
                            pa=Planes(self.sVt.hi[0], self.sVt.lo[0]), ldpa=Np, nsplit=ns, launch=not grp)
            if grp:
                hip.linear_grouped([qk, vs])
            self._mha(self.sQK, 512, 0, self.sQK, 256, 512, 0, self.sVt, None, 1, None)
            hip.linear(self.ao_pl, W.Wp_s, N, 256, 256, ldx=256, ldy=256, bias=W.bp_s, d0=self.tgt.data_ptr(),
                       ldd0=256, accumulate=True, nsplit=ns)
            # -- long / short term (transformer.py:569-592)
            self._ln(self.tgt, W.norm2, self.y_pl, of32=self.y_f32[l])
            pq = hip.linear(self.y_pl, W.Wq, N, 256, 256, ldx=256, ldy=256, bias=W.bq, d0=self.Qc.data_ptr(), ldd0=256,
                            pa=curK, ldpa=256, pb=self.Qpe, ldpb=256, addvec=self.cur_pe, nsplit=ns, launch=not grp)
            pe = W.pe_h[T]
            pb = hip.linear(self.y_pl, pe[0], N, self.HEADS * T, 256, ldx=256, ldy=256, bias=pe[1],
                            d0=self.bias_h.data_ptr(), ldd0=self.HEADS * T, nsplit=ns, launch=not grp)
            if grp:
                hip.linear_grouped([pq, pb])
            if ref_frame:
                self._add_split(self.y_f32[l], self.idemb, out=self.yid_pl)
                hip.linear(W.Wv, self.yid_pl, 256, N, 256, ldx=256, ldy=256, bias=W.bv, bias_per_row=True,
                           pa=curV, ldpa=Np, nsplit=ns)
                hip.linear(self.yid_pl, W.Wv, N, 256, 256, ldx=256, ldy=256, bias=W.bv, d0=self.refV.data_ptr(),
                           ldd0=256, nsplit=ns)
                local_K, local_V = self.Qc, self.refV
            else:
                local_K, local_V = sK[l], sV[l]
            # (shared launches: the long-term read and the short-term read below are independent -- ONE flash launch and one
            # combine launch for both, after the short-term read's operands are there; not on the frames bench.py times the
            # long-term kernel on)
            pair = grp and not self._timing
            lt = (self.Qpe, 256, 0, self.bankK[l], 0, 256, kss, self.bankV[l], map_bank, T, self.bias_h)
            if not pair:
                self._mha(*lt, want_mass=(l == 0), timed=self._timing)
            if not grp:
                hip.linear(self.ao_pl, W.Wp_lt, N, 256, 256, ldx=256, ldy=256, bias=W.bp_lt, d0=self.tgt.data_ptr(),
                           ldd0=256, accumulate=True, nsplit=ns)
            # -- short term on norm4(local + curr) (transformer.py:656-662): the two norms in one launch
            Ks0 = Planes(self.Ks_pl.hi[0], self.Ks_pl.lo[0])
            if grp:
                self._ln_multi([(local_K, W.norm4, Ks0, self.Qc, None, None),
                                (local_V, W.norm4, self.Vs_pl, self.y_f32[l], None, None)])
            else:
                self._ln(local_K, W.norm4, Ks0, x2=self.Qc)
                self._ln(local_V, W.norm4, self.Vs_pl, x2=self.y_f32[l])
            hip.check(lib.rmem_transpose_planes(self.Vs_pl.hi.data_ptr(), self.Vs_pl.lo.data_ptr(), 256, N, 256,
                                                self.VsT.hi.data_ptr(), self.VsT.lo.data_ptr(), Np,
                                                hip.stream_ptr()), "rmem_transpose_planes")
            ao_st = self.ao2_pl if grp else self.ao_pl
            st_ = (curK, 256, 0, self.Ks_pl, 0, 256, 0, self.VsT, None, 1, None)
            if pair:
                self._mha_pair(self._mha_args(*lt, l == 0), self._mha_args(*st_, False, out=ao_st, ws=1), want_mass=(l == 0))
            else:
                self._mha(*st_, out=ao_st)
            # (grouped: the long-term read's projection -- it does not feed the short-term read -- shares the launch of the
            # short-term one's; tgt receives the same two additions in the same order)
            plt = hip.linear(self.ao_pl, W.Wp_lt, N, 256, 256, ldx=256, ldy=256, bias=W.bp_lt, d0=self.tgt.data_ptr(),
                             ldd0=256, accumulate=True, nsplit=ns, launch=False) if grp else None
            pst = hip.linear(ao_st, W.Wp_st, N, 256, 256, ldx=256, ldy=256, bias=W.bp_st,
                             d0=self.tgt3[l].data_ptr(), ldd0=256, pa=self.t3_pl, ldpa=256, nsplit=ns, launch=not grp)
            if grp:
                hip.linear_grouped([plt, pst])
            if not grp:
                self._add_split(self.tgt, self.tgt3[l], dst=self.tgt)                   # tgt += tgt3 (:680)
            qm = hip.linear(self.t3_pl, W.Wqm, N, 256, 256, ldx=256, ldy=256, bias=W.bqm, d0=nK[l].data_ptr(),
                            ldd0=256, nsplit=ns, launch=not grp)                        # local_K (:675)
            if ref_frame:
                self._add_split(self.tgt3[l], self.idemb, out=self.t3id_pl)
                hip.linear(self.t3id_pl, W.Wvm, N, 256, 256, ldx=256, ldy=256, bias=W.bvm,
                           d0=nV[l].data_ptr(), ldd0=256, nsplit=ns)
            # -- feed forward (transformer.py:683-687, basic.py:15-35); its first projection shares the launch of local_K's
            if grp:                     # tgt += tgt3 (:680) inside the norm's launch: the same single addition per element
                self._ln_multi([(self.tgt, W.norm3, self.z_pl, self.tgt3[l], None, None, self.tgt)])
            else:
                self._ln(self.tgt, W.norm3, self.z_pl)
            ff1 = hip.linear(self.z_pl, W.W1, N, self.FF, 256, ldx=256, ldy=256, bias=W.b1, d0=self.a.data_ptr(),
                             ldd0=self.FF, nsplit=ns, launch=not grp)
            if grp:
                hip.linear_grouped([ff1, qm])
            hip.check(lib.rmem_gn_gelu_tokens(self.a.data_ptr(), N, self.FF, 32, W.gn[0].data_ptr(),
                                              W.gn[1].data_ptr(), 1e-5, self.gn_ws.data_ptr(), self.g.data_ptr(),
                                              hip.stream_ptr()), "rmem_gn_gelu_tokens")
            hip.check(lib.rmem_dwconv5x5_split(self.g.data_ptr(), self.FF, W.dw.data_ptr(), self.h, self.w, self.FF,
                                               self.gdw_pl.hi.data_ptr(), self.gdw_pl.lo.data_ptr(), self.FF,
                                               hip.stream_ptr()), "rmem_dwconv5x5_split")
            hip.linear(self.gdw_pl, W.W2, N, 256, self.FF, ldx=self.FF, ldy=self.FF, bias=W.b2,
                       d0=self.tgt.data_ptr(), ldd0=256, accumulate=True, nsplit=ns)
            if grp and l + 1 < self.L:  # the layer's output norm (:248-259) and the next layer's norm1 pair read the same tgt
                Wn = self.lw[l + 1]
                self._ln_multi([(self.tgt, W.dnorm, None, None, None, self.outs[l]),
                                (self.tgt, Wn.norm1, self.x_pl, None, None, None),
                                (self.tgt, Wn.norm1, self.xq_pl, None, self.pos, None)])
            else:
                self._ln(self.tgt, W.dnorm, None, of32=self.outs[l])                    # :248-259

    # ------------------------------------------------------------------ memory update
    def update_short_memories(self, update_long: bool):
        """update_short_memories + update_long_term_memory (transformer.py:269-322)."""
        self._update_device(update_long)
        self._update_host(update_long)

    def update_key(self, update_long: bool):
        return (self.cur, self._flip, bool(update_long))

    def _update_device(self, update_long: bool):
        N, Np, ns = self.N, self.Npad, self.nsplit
        nV = self.nsV if self._flip == 0 else self.sV
        if self.group_launches:
            # every layer's sums in ONE launch, every layer's projections in ONE (independent problems: two launches per
            # update instead of two or four per layer; per problem the same arithmetic)
            sums, lins = [], []
            for l in range(self.L):
                W = self.lw[l]
                if update_long:      # curr_V <- linear_V(curr_V + id_emb) (only consumed by the long bank)
                    sums.append((self.y_f32[l], self.idemb, None, self.yid_l[l]))
                    lins.append(hip.linear(W.Wv, self.yid_l[l], 256, N, 256, ldx=256, ldy=256, bias=W.bv, bias_per_row=True,
                                           pa=self.bankV[l][self.cur], ldpa=Np, nsplit=ns, launch=False))
                sums.append((self.tgt3[l], self.idemb, None, self.t3id_l[l]))
                lins.append(hip.linear(self.t3id_l[l], W.Wvm, N, 256, 256, ldx=256, ldy=256, bias=W.bvm,
                                       d0=nV[l].data_ptr(), ldd0=256, nsplit=ns, launch=False))
            arr = (hip.AddArgs * len(sums))()
            for q, (a, b, dst, out) in zip(arr, sums):
                q.a, q.b, q.dst, q.oh, q.ol = a.data_ptr(), b.data_ptr(), hip.ptr(dst), out.hi.data_ptr(), out.lo.data_ptr()
            hip.check(hip.load().rmem_add_split_multi(arr, len(sums), N * 256, hip.stream_ptr()), "rmem_add_split_multi")
            for k in range(0, len(lins), 8):
                hip.linear_grouped(lins[k:k + 8])
            return
        for l in range(self.L):
            W = self.lw[l]
            if update_long:      # curr_V <- linear_V(curr_V + id_emb) (only consumed by the long bank)
                self._add_split(self.y_f32[l], self.idemb, out=self.yid_pl)
                hip.linear(W.Wv, self.yid_pl, 256, N, 256, ldx=256, ldy=256, bias=W.bv, bias_per_row=True,
                           pa=self.bankV[l][self.cur], ldpa=Np, nsplit=ns)
            self._add_split(self.tgt3[l], self.idemb, out=self.t3id_pl)
            hip.linear(self.t3id_pl, W.Wvm, N, 256, 256, ldx=256, ldy=256, bias=W.bvm, d0=nV[l].data_ptr(),
                       ldd0=256, nsplit=ns)

    def _update_host(self, update_long: bool, frame_index: int = 0):
        self._swap_short()
        if update_long:
            self.bank = self.bank + [self.cur]

    def restrict_long_memories(self, indexes: List[int], fg: torch.Tensor) -> Optional[int]:
        """LongShortTermTransformer.restrict_long_memories (transformer.py:324-436): early
        return while the bank is within its cap (:332-334)."""
        if len(self.bank) <= self.cap:
            return None
        T = self.mass_T
        hip.check(hip.load().rmem_attn_mass_reduce(self.mass.data_ptr(), self.N, T, fg.data_ptr(),
                                                   self.w_out.data_ptr(), hip.stream_ptr()),
                  "rmem_attn_mass_reduce")
        w = self.w_out[:T].cpu().numpy().astype(np.float32)
        w = w / w.sum(dtype=np.float32)
        drop, self.ema, self.visits = rmem_policy_step(w, indexes, self.ema, self.visits,
                                                       self.cfg.FORMER_MEM_LEN)
        del self.bank[drop]
        indexes.remove(indexes[drop])
        return drop
