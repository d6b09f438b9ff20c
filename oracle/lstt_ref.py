"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the RMem /
DeAOT-GPM hot path in explicit fp32 math (PyTorch-CPU tensors, no nn.Module state).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this package.  The product path (``rmem_amd``) never does.

Parity status: **pinned against outputs of the reference itself** run in the build
container -- ``tests/golden/*.npz`` were produced by ``tests/golden/make_golden.py``
importing /root/reference (the reference has no tests / golden vectors of its own,
SURVEY.md section 4 and 8c), and ``tests/test_oracle_vs_reference.py`` re-checks the
restatement directly against the imported reference whenever /root/reference exists.

Every function cites the reference lines it restates (paths relative to
/root/reference/aot_plus/).  Layout: tokens are row-major p = y*w + x; all tensors are
[N, C] (the reference's [N, 1, C] with the batch axis dropped).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ----------------------------------------------------------------------------- basics
def silu(x: Tensor) -> Tensor:
    """networks/layers/attention.py:89-90."""
    return x * torch.sigmoid(x)


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    """nn.LayerNorm over the channel axis (layers/transformer.py:14-20)."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def linear(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    y = x @ w.t()
    return y if b is None else y + b


def dwconv5x5(x_nc: Tensor, w: Tensor, h: int, wd: int) -> Tensor:
    """DWConv2d (layers/basic.py:38-57): [N,C] -> [1,C,h,w] depth-wise 5x5, pad 2,
    no bias, back to [N,C].  Dropout2d is identity in eval."""
    n, c = x_nc.shape
    x = x_nc.view(h, wd, c).permute(2, 0, 1).unsqueeze(0)
    y = F.conv2d(x, w, None, padding=2, groups=c)
    return y[0].permute(1, 2, 0).reshape(n, c)


def temporal_pe_rows(T: int) -> List[int]:
    """Row of mem_pos_emb used for bank position t (layers/transformer.py:1144-1167).

    T==1 -> row 0; T<=4 -> row t (linear interpolation 4->T of the first T rows with
    align_corners is the identity); T>4 -> linear->4 (identity), flip, nearest->T, flip,
    i.e. row(t) = 3 - floor((T-1-t)*4/T).
    """
    if T <= 4:
        return list(range(T))
    return [3 - ((T - 1 - t) * 4) // T for t in range(T)]


# ----------------------------------------------------------------------------- B2
def gated_propagation_core(Q: Tensor, K: Tensor, V: Tensor, U: Tensor, h: int, w: int,
                           dw_w: Tensor, proj_w: Tensor, proj_b: Tensor,
                           d_att: int = 128) -> Tuple[Tensor, Tensor, Tensor]:
    """GatedPropagation.forward after the optional linears, heads=1
    (layers/attention.py:174-211).  Returns (out [N,512], attn [N,M], logits [N,M])."""
    logits = (Q / (d_att ** 0.5)) @ K.t()
    attn = torch.softmax(logits, dim=-1)
    out = (attn @ V) * U
    out = dwconv5x5(out, dw_w, h, w)
    out = linear(out, proj_w, proj_b)
    return out, attn, logits


def self_gated_propagation(s: Tensor, sd: SD, pfx: str, h: int, w: int) -> Tensor:
    """GatedPropagation with use_linear=True on q=k=v=u=s [N,512]
    (layers/attention.py:151-172,195-211); SDPA scale is 1/sqrt(d_att)."""
    half = s.shape[1] // 2
    QK = linear(s, sd[pfx + "linear_QK.weight"], sd[pfx + "linear_QK.bias"])
    s1, s2 = s[:, :half], s[:, half:]
    V = silu(torch.cat([linear(s1, sd[pfx + "linear_V1.weight"], sd[pfx + "linear_V1.bias"]),
                        linear(s2, sd[pfx + "linear_V2.weight"], sd[pfx + "linear_V2.bias"])], dim=1))
    U = silu(torch.cat([linear(s1, sd[pfx + "linear_U1.weight"], sd[pfx + "linear_U1.bias"]),
                        linear(s2, sd[pfx + "linear_U2.weight"], sd[pfx + "linear_U2.bias"])], dim=1))
    out, _, _ = gated_propagation_core(QK, QK, V, U, h, w, sd[pfx + "dw_conv.conv.weight"],
                                       sd[pfx + "projection.weight"], sd[pfx + "projection.bias"],
                                       d_att=QK.shape[1])
    return out


# ----------------------------------------------------------------------------- B3
def local_window_index(h: int, w: int, max_dis: int = 7):
    """For every query p and window offset o: key index (or -1 if outside the image).
    Restates the geometry of pad_and_unfold / qk_mask (layers/attention.py:305-312,
    404-413): offset o = (dy+7)*15 + (dx+7), key = (y+dy, x+dx)."""
    win = 2 * max_dis + 1
    ys = torch.arange(h).view(h, 1, 1, 1)
    xs = torch.arange(w).view(1, w, 1, 1)
    dy = torch.arange(-max_dis, max_dis + 1).view(1, 1, win, 1)
    dx = torch.arange(-max_dis, max_dis + 1).view(1, 1, 1, win)
    ky, kx = ys + dy, xs + dx
    inside = (ky >= 0) & (ky < h) & (kx >= 0) & (kx < w)
    idx = torch.where(inside, ky * w + kx, torch.full_like(ky * w + kx, -1))
    return idx.reshape(h * w, win * win), inside.reshape(h * w, win * win)


def local_gated_propagation(q: Tensor, k: Tensor, v: Tensor, u: Tensor, h: int, w: int,
                            rel_w: Tensor, rel_b: Tensor, dw_w: Tensor, proj_w: Tensor,
                            proj_b: Tensor, max_dis: int = 7, logits_out: Optional[list] = None) -> Tuple[Tensor, Tensor]:
    """LocalGatedPropagation.forward, use_linear=False, enable_corr=False, heads=1
    (layers/attention.py:289-361).  q,k [N,128] (unscaled), v [N,1024], u [N,1024].
    Returns (out [N,512], local_attn [N,225]); logits_out (a list) receives the pre-softmax logits [N,225]."""
    n, d = q.shape
    idx, inside = local_window_index(h, w, max_dis)
    rel = q @ rel_w.view(rel_w.shape[0], d).t() + rel_b          # :314 (unscaled q)
    qs = q / (d ** 0.5)                                          # :317
    kg = k[idx.clamp(min=0)]                                     # [N,225,128]; zero pad == masked anyway
    kg = kg * inside.unsqueeze(-1)
    qk = torch.einsum("nc,noc->no", qs, kg) + rel                # :334-342
    qk = qk - (~inside).float() * 1e8                            # :344
    if logits_out is not None:
        logits_out.append(qk)
    attn = torch.softmax(qk, dim=1)                              # :346
    vg = v[idx.clamp(min=0)] * inside.unsqueeze(-1)              # local2global + matmul, :350-353
    agg = torch.einsum("no,noc->nc", attn, vg)
    out = agg * u                                                # :355
    out = dwconv5x5(out, dw_w, h, w)                             # :357
    out = linear(out, proj_w, proj_b)                            # :358
    return out, attn


# ----------------------------------------------------------------------------- B1
class Memory:
    """Per-layer memory state: bank slots (PE-free) and the short-term frame."""

    def __init__(self):
        self.K: List[Tensor] = []      # per slot [N,128]
        self.V: List[Tensor] = []      # per slot [N,512]
        self.IDV: List[Tensor] = []    # per slot [N,512]
        self.sK: Optional[Tensor] = None
        self.sV: Optional[Tensor] = None
        self.sIDV: Optional[Tensor] = None


def fuse_key_value_id(sd: SD, layer: int, z: Optional[Tensor], id_emb: Tensor) -> Tensor:
    """GatedPropagationModule.fuse_key_value_id (layers/transformer.py:1238-1244)."""
    p = f"LSTT.layers.{layer}."
    x = id_emb if z is None else torch.cat([z, id_emb], dim=1)
    return silu(linear(x, sd[p + "linear_ID_V.weight"], sd[p + "linear_ID_V.bias"]))


def gpm_layer(sd: SD, layer: int, tgt: Tensor, tgt_id: Optional[Tensor], mem: Memory,
              h: int, w: int, cur_pe: Tensor, mem_pe: Tensor,
              curr_id_emb: Optional[Tensor] = None, trace: Optional[dict] = None):
    """GatedPropagationModule.forward (layers/transformer.py:1091-1236).

    Returns (tgt, tgt_id, curr=(K,V,z), mass [N,T], (bank_K,bank_V,bank_IDV), (sK,sV,sIDV)).
    ``mass`` is record_attn_weight (:1186-1192)."""
    p = f"LSTT.layers.{layer}."
    d = tgt.shape[1]
    d_att = d // 2
    x = layer_norm(tgt, sd[p + "norm1.weight"], sd[p + "norm1.bias"])            # :1104
    QV = linear(x, sd[p + "linear_QV.weight"], sd[p + "linear_QV.bias"])         # :1106
    Q = QV[:, :d_att]
    V = silu(QV[:, d_att:])                                                      # :1111
    Uraw = linear(x, sd[p + "linear_U.weight"], sd[p + "linear_U.bias"])         # :1112
    if tgt_id is None:                                                           # :1114-1118
        Ucat = torch.cat([silu(Uraw), torch.ones_like(Uraw)], dim=1)
        z = None
        tgt_id_in = torch.zeros_like(tgt)
    else:                                                                        # :1119-1123
        z = layer_norm(tgt_id, sd[p + "id_norm1.weight"], sd[p + "id_norm1.bias"])
        IDU = linear(z, sd[p + "linear_ID_U.weight"], sd[p + "linear_ID_U.bias"])
        Ucat = silu(torch.cat([Uraw, IDU], dim=1))
        tgt_id_in = tgt_id

    if curr_id_emb is not None:                                                  # :1125-1135
        idv = fuse_key_value_id(sd, layer, z, curr_id_emb)
        bank_K, bank_V, bank_IDV = [Q], [V], [idv]
        sK, sV, sIDV = Q, V, idv
    else:                                                                        # :1137-1138
        bank_K, bank_V, bank_IDV = mem.K, mem.V, mem.IDV
        sK, sV, sIDV = mem.sK, mem.sV, mem.sIDV

    T = len(bank_K)
    rows = temporal_pe_rows(T)                                                   # :1140-1172
    Kpe = torch.cat([bank_K[t] + mem_pe[rows[t]].view(1, -1) for t in range(T)], dim=0)
    Qpe = Q + cur_pe.view(1, -1)
    Vcat = torch.cat([torch.cat(bank_V, dim=0), torch.cat(bank_IDV, dim=0)], dim=1)  # :1177-1180

    lp = p + "long_term_attn."
    o2, attn, logits = gated_propagation_core(Qpe, Kpe, Vcat, Ucat, h, w,        # :1183
                                              sd[lp + "dw_conv.conv.weight"],
                                              sd[lp + "projection.weight"],
                                              sd[lp + "projection.bias"], d_att)
    n = tgt.shape[0]
    mass = attn.view(n, T, n).sum(dim=2)                                         # :1186-1192

    sp = p + "short_term_attn."
    o3, local_attn = local_gated_propagation(                                    # :1199
        Q, sK, torch.cat([sV, sIDV], dim=1), Ucat, h, w,
        sd[sp + "relative_emb_k.weight"], sd[sp + "relative_emb_k.bias"],
        sd[sp + "dw_conv.conv.weight"], sd[sp + "projection.weight"],
        sd[sp + "projection.bias"], logits_out=(st_logits := []))

    tgt = tgt + o2[:, :d] + o3[:, :d]                                            # :1212-1220
    tgt_id = tgt_id_in + o2[:, d:] + o3[:, d:]

    s = torch.cat([layer_norm(tgt, sd[p + "norm2.weight"], sd[p + "norm2.bias"]),     # :1223-1225
                   layer_norm(tgt_id, sd[p + "id_norm2.weight"], sd[p + "id_norm2.bias"])], dim=1)
    o = self_gated_propagation(s, sd, p + "self_attn.", h, w)                    # :1227
    tgt = tgt + o[:, :d]                                                         # :1231-1232
    tgt_id = tgt_id + o[:, d:]
    if trace is not None:
        trace[f"l{layer}.Q"] = Q
        trace[f"l{layer}.V"] = V
        trace[f"l{layer}.Ucat"] = Ucat
        trace[f"l{layer}.lt_logits"] = logits
        trace[f"l{layer}.st_logits"] = st_logits[0]
        trace[f"l{layer}.o2"] = o2
        trace[f"l{layer}.o3"] = o3
        trace[f"l{layer}.local_attn"] = local_attn
        trace[f"l{layer}.mass"] = mass
        trace[f"l{layer}.tgt"] = tgt
        trace[f"l{layer}.tgt_id"] = tgt_id
    return tgt, tgt_id, (Q, V, z), mass, (bank_K, bank_V, bank_IDV), (sK, sV, sIDV)


def group_norm_tokens(x: Tensor, w: Tensor, b: Tensor, groups: int = 2, eps: float = 1e-5) -> Tensor:
    """GroupNorm1D over a [N,C] sequence (layers/basic.py:6-12): statistics over
    (C/groups channels x N tokens) per group."""
    n, c = x.shape
    xg = x.t().reshape(groups, -1)
    mu = xg.mean(dim=1, keepdim=True)
    var = ((xg - mu) ** 2).mean(dim=1, keepdim=True)
    y = ((xg - mu) / torch.sqrt(var + eps)).reshape(c, n).t()
    return y * w + b


class DeAOTOracle:
    """DualBranchGPM state + forward/update/restrict (layers/transformer.py:700-1007)."""

    def __init__(self, sd: SD, num_layers: int = 3):
        self.sd = sd
        self.L = num_layers
        self.clear_memory()

    def clear_memory(self):                                                      # :1000-1007
        self.mem = [Memory() for _ in range(self.L)]
        self.curr = None
        self.mass0 = None
        self.ema: Dict[int, float] = {}
        self.visits: Dict[int, int] = {}
        self._pending = None

    def forward(self, emb: Tensor, h: int, w: int, curr_id_emb: Optional[Tensor] = None,
                trace: Optional[dict] = None) -> Tensor:
        """DualBranchGPM.forward (:765-824) + final GroupNorm (:806-808).  Returns [N,512]."""
        sd = self.sd
        cur_pe, mem_pe = sd["cur_pos_emb"][0], sd["mem_pos_emb"]
        tgt, tgt_id = emb, None
        curr, pending = [], []
        for l in range(self.L):
            tgt, tgt_id, c, mass, bank, short = gpm_layer(
                sd, l, tgt, tgt_id, self.mem[l], h, w, cur_pe, mem_pe, curr_id_emb, trace)
            curr.append(c)
            pending.append((bank, short))
            if l == 0:
                self.mass0 = mass
        self.curr = curr
        self._pending = pending
        out = torch.cat([tgt, tgt_id], dim=1)
        return group_norm_tokens(out, sd["LSTT.decoder_norms.0.gn.weight"],
                                 sd["LSTT.decoder_norms.0.gn.bias"], 2)

    def init_memory(self):
        """DualBranchGPM.init_memory (:993-998): bank := the reference frame."""
        for l in range(self.L):
            (bK, bV, bIDV), (sK, sV, sIDV) = self._pending[l]
            m = self.mem[l]
            m.K, m.V, m.IDV = list(bK), list(bV), list(bIDV)
            m.sK, m.sV, m.sIDV = sK, sV, sIDV
        self.ema, self.visits = {}, {}

    def update_short_memories(self, id_emb: Tensor, update_long: bool):
        """update_short_memories + update_long_term_memory (:826-878)."""
        for l in range(self.L):
            K, V, z = self.curr[l]
            idv = fuse_key_value_id(self.sd, l, z, id_emb)
            m = self.mem[l]
            m.sK, m.sV, m.sIDV = K, V, idv
            if update_long:
                m.K = m.K + [K]
                m.V = m.V + [V]
                m.IDV = m.IDV + [idv]

    def restrict_long_memories(self, former: int, latter: int, indexes: List[int],
                               fg: Tensor, log: Optional[dict] = None) -> Optional[int]:
        """DualBranchGPM.restrict_long_memories (:880-991), use_atten_weight=True.

        ``indexes`` is mutated like the reference's long_memories_indexes.  Returns the
        dropped bank position or None."""
        mass = self.mass0                                                        # :894-899
        wgt = (mass * fg.reshape(-1, 1)).sum(dim=0)                              # :900-904
        wgt = wgt / wgt.sum()                                                    # :905
        drop, ema, visits, scores = rmem_policy_step(
            [float(x) for x in wgt], indexes, self.ema, self.visits, former)
        self.ema, self.visits = ema, visits
        if log is not None:
            log.update(w=[float(x) for x in wgt], scores=scores, drop=drop,
                       ema=dict(ema), visits=dict(visits))
        cap = former + latter
        dropped = None
        for l in range(self.L):                                                  # :967-989
            m = self.mem[l]
            if len(m.K) > cap:
                dropped = drop
                del m.K[drop], m.V[drop], m.IDV[drop]
        if dropped is not None:
            indexes.remove(indexes[drop])                                        # :990-991
        return dropped


def rmem_policy_step(w: List[float], indexes: List[int], ema_prev: Dict[int, float],
                     visits_prev: Dict[int, int], former: int):
    """The EMA + UCB eviction rule (layers/transformer.py:909-964) in float32 arithmetic.

    ``w`` has one entry per attended slot (= len(indexes)-1, the newest appended slot
    was not attended yet).  Returns (drop_position, new_ema, new_visits, scores)."""
    f32 = torch.float32
    wv = torch.tensor(w, dtype=f32)
    n_att = len(w)
    ema = {}
    for i in range(n_att):                                                       # :910-923
        idx = indexes[i]
        if idx in ema_prev:
            val = (1 - 0.8) * ema_prev[idx] + 0.8 * wv[i]
        else:
            val = wv[i]
        ema[idx] = val
    for i in range(n_att):                                                       # :926-927
        wv[i] = ema[indexes[i]]
    visits = {idx: (1 + visits_prev[idx]) if idx in visits_prev else 1 for idx in indexes}  # :931-941
    c = torch.tensor([float(visits[idx]) for idx in indexes[:-1]], dtype=f32)    # :942-945
    c[0] = len(c)                                                                # :947
    bonus = 1.5 * torch.sqrt(torch.log(c.sum()) / (c + 8))                       # :950-952
    score = wv + bonus                                                           # :954
    drop = former                                                                # :888
    if score.numel() > 1:                                                        # :958-964
        drop = int(torch.argmin(score[1:]).item()) + 1
    return drop, ema, visits, [float(s) for s in score]


# ----------------------------------------------------------------------------- B5
def id_assign(label: Tensor, sd: SD, max_obj: int = 10, deaot: bool = True,
              stride: int = 16, ksize: int = 0, pad: int = -1, use_ignore: bool = True) -> Tensor:
    """one_hot_mask + assign_identity + get_id_emb
    (utils/image.py:69-74, engines/aot_engine.py:208-232, models/deaot.py:65-69,
    models/aot.py:67-74,111-114).  label [1,1,H,W] float ids (255 = ignore) -> [N,256].
    use_ignore=True is update_short_term_memory (aot_engine.py:330-336: the (mask == 255) map
    is the ignore channel); use_ignore=False is add_reference_frame (:304), which calls
    assign_identity without an ignore mask, i.e. with zeros (:209-213): a 255 pixel then has
    neither a one-hot nor an ignore channel."""
    if ksize == 0:      # k17/p8 with MODEL_ALIGN_CORNERS, k16/p0 without (models/aot.py:67-84)
        ksize = sd["patch_wise_id_bank.weight"].shape[-1]
        pad = 8 if ksize == 17 else 0
    ids = torch.arange(0, max_obj + 1).view(1, -1, 1, 1).to(label.dtype)
    onehot = (label == ids).float()
    ign = (label == 255).float() if use_ignore else torch.zeros_like(label)
    onehot[:, 0] = onehot[:, 0] * (ign[:, 0] == 0).float()
    x = torch.cat([onehot, ign], dim=1)
    e = F.conv2d(x, sd["patch_wise_id_bank.weight"], sd["patch_wise_id_bank.bias"],
                 stride=stride, padding=pad)
    e = e[0].permute(1, 2, 0).reshape(-1, e.shape[1])
    if deaot:
        e = layer_norm(e, sd["id_norm.weight"], sd["id_norm.bias"])
    return e


def foreground_proba(pred_id_logits: Tensor, h: int, w: int) -> Tensor:
    """engines/aot_engine.py:355-362: 1 - softmax(bilinear(align_corners=True))[:,0]."""
    lg = F.interpolate(pred_id_logits, size=(h, w), mode="bilinear", align_corners=True)
    return 1 - torch.softmax(lg, dim=1)[:, 0:1]
