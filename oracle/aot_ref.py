"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the AOT variant of
the hot path: ``SimplifiedTransformerBlock`` / ``LongShortTermTransformer`` with 8-head
``MultiheadAttention``, explicit fp32 math on PyTorch-CPU.

Parity status: pinned against outputs of the reference itself (tests/golden/aot_*.npz,
clip_aot_*.json produced by tests/golden/make_golden.py from /root/reference).
Paths cited are relative to /root/reference/aot_plus/.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from .lstt_ref import (dwconv5x5, layer_norm, linear, rmem_policy_step, temporal_pe_rows)

Tensor = torch.Tensor
SD = Dict[str, Tensor]


def sine_pos_emb(h: int, w: int, num_pos_feats: int = 128, temperature: float = 10000.0) -> Tensor:
    """PositionEmbeddingSine(normalize=True) (networks/layers/position.py:35-77) -> [N, 256]
    token-major (channels: 128 y-features then 128 x-features)."""
    scale = 2 * math.pi
    eps = 1e-6
    ys = torch.arange(h, dtype=torch.float32).view(h, 1).expand(h, w)
    xs = torch.arange(w, dtype=torch.float32).view(1, w).expand(h, w)
    y_embed = ys / (ys[-1:, :] + eps) * scale
    x_embed = xs / (xs[:, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="trunc") / num_pos_feats)
    pos_x = x_embed[:, :, None] / dim_t
    pos_y = y_embed[:, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, 0::2].sin(), pos_x[:, :, 1::2].cos()), dim=3).flatten(2)
    pos_y = torch.stack((pos_y[:, :, 0::2].sin(), pos_y[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((pos_y, pos_x), dim=2).reshape(h * w, 2 * num_pos_feats)


def mha_core(Q: Tensor, K: Tensor, V: Tensor, heads: int = 8):
    """MultiheadAttention after the optional linears (layers/attention.py:45-77):
    per head softmax((Q/sqrt(d_h)) K^T) V.  Returns (out [N,256], attn [heads,N,M])."""
    n, d = Q.shape
    dh = d // heads
    q = (Q / (dh ** 0.5)).view(n, heads, dh).permute(1, 0, 2)
    k = K.view(-1, heads, dh).permute(1, 2, 0)
    v = V.view(-1, heads, dh).permute(1, 0, 2)
    attn = torch.softmax(q @ k, dim=-1)
    out = (attn @ v).permute(1, 0, 2).reshape(n, d)
    return out, attn


class AOTMemory:
    def __init__(self):
        self.K: List[Tensor] = []      # per slot [N,256] (PE-free)
        self.V: List[Tensor] = []      # per slot [N,256]
        self.sK: Optional[Tensor] = None
        self.sV: Optional[Tensor] = None


def aot_block(sd: SD, layer: int, tgt: Tensor, mem: AOTMemory, h: int, w: int, pos: Tensor,
              cur_pe: Tensor, mem_pe: Tensor, curr_id_emb: Optional[Tensor] = None,
              trace: Optional[dict] = None):
    """SimplifiedTransformerBlock.forward, linear_q=False (layers/transformer.py:553-692).

    Returns (tgt, curr=(Kc, Vc), mass [N,T], bank=(K list, V list), short=(local_K, local_V))."""
    p = f"LSTT.layers.{layer}."
    W = lambda k: sd[p + k]
    # -- self attention (:558-566)
    x = layer_norm(tgt, W("norm1.weight"), W("norm1.bias"))
    qk = x + pos
    sp = p + "self_attn."
    o, _ = mha_core(linear(qk, sd[sp + "linear_Q.weight"], sd[sp + "linear_Q.bias"]),
                    linear(qk, sd[sp + "linear_K.weight"], sd[sp + "linear_K.bias"]),
                    linear(x, sd[sp + "linear_V.weight"], sd[sp + "linear_V.bias"]))
    tgt = tgt + linear(o, sd[sp + "projection.weight"], sd[sp + "projection.bias"])
    # -- long / short term (:569-592)
    y = layer_norm(tgt, W("norm2.weight"), W("norm2.bias"))
    Qc = linear(y, W("linear_Q.weight"), W("linear_Q.bias"))
    Kc, Vc = Qc, y
    if curr_id_emb is not None:
        gV = linear(Vc + curr_id_emb, W("linear_V.weight"), W("linear_V.bias"))
        bank_K, bank_V = [Kc], [gV]
        local_K, local_V = Kc, gV
    else:
        bank_K, bank_V = mem.K, mem.V
        local_K, local_V = mem.sK, mem.sV
    T = len(bank_K)
    rows = temporal_pe_rows(T)                                                   # :594-629
    Kpe = torch.cat([bank_K[t] + mem_pe[rows[t]].view(1, -1) for t in range(T)], dim=0)
    Qpe = Qc + cur_pe.view(1, -1)
    o2, attn = mha_core(Qpe, Kpe, torch.cat(bank_V, dim=0))                      # :632-635
    lp = p + "long_term_attn."
    tgt2 = linear(o2, sd[lp + "projection.weight"], sd[lp + "projection.bias"])
    n = tgt.shape[0]
    mass = attn.mean(dim=0).view(n, T, n).sum(dim=2)                             # :636-644
    # -- short term, norm4 variant (:656-662)
    Ks = layer_norm(local_K + Kc, W("norm4.weight"), W("norm4.bias"))
    Vs = layer_norm(local_V + Vc, W("norm4.weight"), W("norm4.bias"))
    o3, _ = mha_core(Qc, Ks, Vs)
    stp = p + "short_term_attn."
    tgt3 = linear(o3, sd[stp + "projection.weight"], sd[stp + "projection.bias"])
    new_local_K = linear(tgt3, W("linear_QMem.weight"), W("linear_QMem.bias"))  # :675-678
    new_local_V = tgt3
    if curr_id_emb is not None:
        new_local_V = linear(tgt3 + curr_id_emb, W("linear_VMem.weight"), W("linear_VMem.bias"))
    tgt = tgt + tgt2 + tgt3                                                      # :680
    # -- feed forward (:683-687): linear1 -> GN(32)+GELU+DW5x5 (basic.py:15-35) -> linear2
    z = layer_norm(tgt, W("norm3.weight"), W("norm3.bias"))
    a = linear(z, W("linear1.weight"), W("linear1.bias"))                        # [N,1024]
    a4 = a.view(h, w, -1).permute(2, 0, 1).unsqueeze(0)
    a4 = F.gelu(F.group_norm(a4, 32, W("activation.gn.weight"), W("activation.gn.bias"), 1e-5))
    a = a4[0].permute(1, 2, 0).reshape(n, -1)
    a = dwconv5x5(a, W("activation.conv.weight"), h, w)
    tgt = tgt + linear(a, W("linear2.weight"), W("linear2.bias"))
    if trace is not None:
        trace[f"l{layer}.tgt"] = tgt
        trace[f"l{layer}.tgt3"] = tgt3
        trace[f"l{layer}.mass"] = mass
    return tgt, (Kc, Vc), mass, (bank_K, bank_V), (new_local_K, new_local_V)


class AOTOracle:
    """LongShortTermTransformer state + forward/update/restrict
    (layers/transformer.py:133-464)."""

    def __init__(self, sd: SD, num_layers: int = 3):
        self.sd = sd
        self.L = num_layers
        self.clear_memory()

    def clear_memory(self):
        self.mem = [AOTMemory() for _ in range(self.L)]
        self.curr = None
        self.short_next = None
        self.mass0 = None
        self.ema: Dict[int, float] = {}
        self.visits: Dict[int, int] = {}
        self._pending = None

    def forward(self, emb: Tensor, h: int, w: int, pos: Tensor, curr_id_emb: Optional[Tensor] = None,
                trace: Optional[dict] = None) -> List[Tensor]:
        """Returns the 3 per-layer outputs after their LayerNorms (:248-259)."""
        sd = self.sd
        cur_pe, mem_pe = sd["cur_pos_emb"][0], sd["mem_pos_emb"]
        tgt = emb
        outs, curr, pending, short = [], [], [], []
        for l in range(self.L):
            tgt, c, mass, bank, sh = aot_block(sd, l, tgt, self.mem[l], h, w, pos, cur_pe, mem_pe,
                                               curr_id_emb, trace)
            outs.append(layer_norm(tgt, sd[f"LSTT.decoder_norms.{l}.weight"],
                                   sd[f"LSTT.decoder_norms.{l}.bias"]))
            curr.append(list(c))
            pending.append(bank)
            short.append(list(sh))
            if l == 0:
                self.mass0 = mass
        self.curr, self._pending, self.short_next = curr, pending, short
        return outs

    def init_memory(self):                                                       # :438-453
        for l in range(self.L):
            bK, bV = self._pending[l]
            m = self.mem[l]
            m.K, m.V = list(bK), list(bV)
            m.sK, m.sV = self.short_next[l]
        self.ema, self.visits = {}, {}

    def update_short_memories(self, id_emb: Tensor, update_long: bool):         # :269-322
        sd = self.sd
        for l in range(self.L):
            p = f"LSTT.layers.{l}."
            Kc, Vc = self.curr[l]
            Vl = linear(Vc + id_emb, sd[p + "linear_V.weight"], sd[p + "linear_V.bias"])
            sK, sV = self.short_next[l]
            sV = linear(sV + id_emb, sd[p + "linear_VMem.weight"], sd[p + "linear_VMem.bias"])
            m = self.mem[l]
            m.sK, m.sV = sK, sV
            if update_long:
                m.K = m.K + [Kc]
                m.V = m.V + [Vl]

    def restrict_long_memories(self, former: int, latter: int, indexes: List[int], fg: Tensor,
                               log: Optional[dict] = None) -> Optional[int]:
        """:324-436 -- identical rule to DeAOT but with the early return (:332-334)."""
        cap = former + latter
        if len(self.mem[0].K) <= cap:
            return None
        wgt = (self.mass0 * fg.reshape(-1, 1)).sum(dim=0)
        wgt = wgt / wgt.sum()
        drop, ema, visits, scores = rmem_policy_step([float(x) for x in wgt], indexes, self.ema,
                                                     self.visits, former)
        self.ema, self.visits = ema, visits
        if log is not None:
            log.update(w=[float(x) for x in wgt], drop=drop)
        for l in range(self.L):
            m = self.mem[l]
            del m.K[drop], m.V[drop]
        indexes.remove(indexes[drop])
        return drop


def aot_id_assign(label: Tensor, sd: SD, max_obj: int = 10, use_ignore: bool = True) -> Tensor:
    """AOT.get_id_emb has no LayerNorm (models/aot.py:111-114)."""
    from .lstt_ref import id_assign
    return id_assign(label, sd, max_obj, deaot=False, use_ignore=use_ignore)
