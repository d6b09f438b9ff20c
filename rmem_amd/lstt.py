"""DeAOT LSTT (DualBranchGPM) executor on the HIP kernels of librmem_hip.so.

Host-side mirror of the reference's ``DualBranchGPM`` / ``GatedPropagationModule``
(/root/reference/aot_plus/networks/layers/transformer.py:700-1249): same operations,
same state (long-term bank, short-term frame, EMA / visit dictionaries), but

  * the memory bank is a pre-allocated ring of ``cap + 2`` physical slots per layer
    (K planes [slot][Npad][128], V planes "blocked-16" [slot][Npad/16][1024][16], the
    operand layout of the fused read, csrc/read64.hip); append / evict only edit a
    logical->physical map, no ``torch.cat`` re-allocation (transformer.py:859-878, :967-989);
  * every memory read (long-term bank, windowed short-term, self) is ONE flash-style launch
    (rmem_attn_read / rmem_attn_read2) + one combine: no probability matrix in HBM;
  * stored keys are PE-free, the temporal positional embedding enters the logits as a
    per-(query, slot) bias (transformer.py:1140-1172);
  * there is no D2H sync inside a frame; the RMem relevance vector (<= 16 floats) is
    read back once per long-memory update (transformer.py:906).

PyTorch here is only device memory + streams; every FLOP of the LSTT runs in
rmem_amd/csrc.  There is no CPU fallback.
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


def temporal_pe_rows(T: int) -> List[int]:
    """mem_pos_emb row for bank position t (transformer.py:1144-1167):
    T<=4 -> t; T>4 -> 3 - floor((T-1-t)*4/T)."""
    if T <= 4:
        return list(range(T))
    return [3 - ((T - 1 - t) * 4) // T for t in range(T)]


def rmem_policy_step(w: np.ndarray, indexes: List[int], ema_prev: Dict[int, float],
                     visits_prev: Dict[int, int], former: int):
    """EMA(0.8) + UCB eviction rule in float32 (transformer.py:909-964).

    ``w``: normalised fg-weighted attention mass of the attended slots
    (len(indexes) - 1 entries; the slot appended this frame was not attended).
    Returns (drop_position, ema, visits)."""
    f = np.float32
    n_att = len(w)
    ema: Dict[int, float] = {}
    wv = np.array(w, dtype=f)
    for i in range(n_att):
        idx = indexes[i]
        if idx in ema_prev:
            wv[i] = f(f(1 - 0.8) * f(ema_prev[idx])) + f(f(0.8) * wv[i])
        ema[idx] = f(wv[i])
    visits = {idx: (1 + visits_prev[idx]) if idx in visits_prev else 1 for idx in indexes}
    c = np.array([visits[idx] for idx in indexes[:-1]], dtype=f)
    c[0] = f(len(c))
    bonus = f(1.5) * np.sqrt(np.log(c.sum(dtype=f)) / (c + f(8)))
    score = wv + bonus.astype(f)
    drop = former
    if score.size > 1:
        drop = int(np.argmin(score[1:])) + 1
    return drop, ema, visits


class _LayerWeights:
    pass


class _AttnWS:
    """Workspace of one fused read: split partials, (max, sum) statistics, per-slot sums, G."""

    def __init__(self, T, N, Np, ksplits, dev):
        self.part = torch.zeros(ksplits, Np, 1024, device=dev)
        self.ml = torch.zeros(ksplits, Np, 2, device=dev)
        self.lslot = torch.zeros(ksplits, Np, T, 2, device=dev)
        self.G = torch.zeros(N, 1024, device=dev)


class DeAOTLSTT:
    """HIP-backed DualBranchGPM for one clip geometry (h x w tokens)."""

    D = 256
    DATT = 128
    E = 512          # expanded width per branch
    WIN = 225

    def __init__(self, model, h: int, w: int, device, nsplit: int = 3, clips_per_launch: int = 1,
                 weights_from: "Optional[DeAOTLSTT]" = None):
        """clips_per_launch > 1: this LSTT is one of that many clips served by shared launches
        (rmem_amd.batched): the key splits of the fused reads are sized for the whole launch.
        weights_from: another LSTT of the same model whose packed weights are shared."""
        hip.load()
        self.clips_per_launch = max(1, int(clips_per_launch))
        self._batched = False              # set by rmem_amd.batched: tgt / tgt_id / out are slices of shared buffers
        self.cfg = model.cfg
        self.h, self.w = int(h), int(w)
        self.N = self.h * self.w
        self.Npad = (self.N + 127) // 128 * 128
        self.dev = torch.device(device)
        self.L = self.cfg.MODEL_LSTT_NUM
        self.cap = self.cfg.FORMER_MEM_LEN + self.cfg.LATTER_MEM_LEN
        self.Tmax = self.cap + 1          # bank momentarily holds cap+1 before eviction
        if self.Tmax > 16:
            raise hip.RmemError("bank of more than 15 slots is not supported (temporal-PE rows array)")
        # RMem eviction on the device (csrc/pointwise.hip: rmem_fg_weights / rmem_bank_*): the host learns which slot
        # was dropped from a pinned result word when it next looks (normally one frame later, without waiting), so a
        # frame may start while the bank momentarily counts cap + 1 candidates: one more physical slot than the
        # cap + 2 the host-side rule needs.  RMEM_HOST_POLICY=1 (and the several-clips-per-launch path, the AOT
        # block) keep the rule on the host: one blocking D2H of <= 16 floats per long-memory update.
        self.device_policy = self.clips_per_launch == 1 and os.environ.get("RMEM_HOST_POLICY") != "1"
        self.S = self.cap + 3             # physical slots
        self.nsplit = int(nsplit)
        self._timing = False
        self._events = []
        self._force_tiles = os.environ.get("RMEM_LINEAR", "") == "tiles"
        self._sample_read = False          # bench.py: time the next replayed frame's layer-0 read (DeAOTEngine._graphed_frame)
        self._sample_kernels = False       # bench.py: issue the next frame's `rest` part eagerly with HIP events around every launch
        self.fuse_gate = os.environ.get("RMEM_FUSE_GATE", "1") != "0"      # single-split reads gate their own output (no combine)
        self._kev = None                   # = _kev_store while a sampled frame is being issued, else None
        self._kev_store: list = []         # [(kernel class, e0, e1, algorithmic flops, algorithmic bytes)] of the sampled frames
        self._kev_frames = 0
        self._skip_read2 = False
        self._split_parts = False
        self.scale = 1.0 / math.sqrt(self.DATT)
        if weights_from is not None:
            for k in ("cur_pe", "mem_pe", "id_ksize", "id_ncls", "id_stride", "id_pad", "id_wt", "id_bias", "id_gamma",
                      "id_beta", "gn_gamma", "gn_beta", "lw"):
                setattr(self, k, getattr(weights_from, k))
        else:
            self._pack_weights(model)
        self._alloc()
        self.clear_memory()

    # ------------------------------------------------------------------ weights
    def _pl(self, t: torch.Tensor) -> Planes:
        return Planes.from_f32(t.detach().to(self.dev, torch.float32).contiguous())

    def _f(self, t: torch.Tensor) -> torch.Tensor:
        return t.detach().to(self.dev, torch.float32).contiguous()

    def _pack_weights(self, model):
        sd = model.state_dict()
        self.cur_pe = self._f(sd["cur_pos_emb"][0])
        self.mem_pe = self._f(sd["mem_pos_emb"])
        kb = sd["patch_wise_id_bank.weight"]                      # [256, ncls, k, k]
        self.id_ksize = kb.shape[-1]
        self.id_ncls = kb.shape[1]
        self.id_stride, self.id_pad = (16, 8) if self.cfg.MODEL_ALIGN_CORNERS else (16, 0)
        self.id_wt = self._f(kb.permute(1, 2, 3, 0))               # [ncls][k][k][256]
        self.id_bias = self._f(sd["patch_wise_id_bank.bias"])
        self.id_gamma, self.id_beta = self._f(sd["id_norm.weight"]), self._f(sd["id_norm.bias"])
        self.gn_gamma = self._f(sd["LSTT.decoder_norms.0.gn.weight"])
        self.gn_beta = self._f(sd["LSTT.decoder_norms.0.gn.bias"])
        self.lw = []
        for l in range(self.L):
            p = f"LSTT.layers.{l}."
            W = _LayerWeights()
            g = lambda k: sd[p + k]
            W.ln1 = (self._f(g("norm1.weight")), self._f(g("norm1.bias")))
            W.ln2 = (self._f(g("norm2.weight")), self._f(g("norm2.bias")))
            W.lnid2 = (self._f(g("id_norm2.weight")), self._f(g("id_norm2.bias")))
            qv_w, qv_b = g("linear_QV.weight"), g("linear_QV.bias")
            W.Wq, W.bq = self._pl(qv_w[:self.DATT]), self._f(qv_b[:self.DATT])
            W.Wv, W.bv = self._pl(qv_w[self.DATT:]), self._f(qv_b[self.DATT:])
            W.Wu, W.bu = self._pl(g("linear_U.weight")), self._f(g("linear_U.bias"))
            W.Widv, W.bidv = self._pl(g("linear_ID_V.weight")), self._f(g("linear_ID_V.bias"))
            if l > 0:
                W.lnid1 = (self._f(g("id_norm1.weight")), self._f(g("id_norm1.bias")))
                W.Widu, W.bidu = self._pl(g("linear_ID_U.weight")), self._f(g("linear_ID_U.bias"))
            dw = lambda k: self._f(g(k).reshape(-1, 25).t())          # [1024,1,5,5] -> [25][1024]
            W.dw_lt, W.dw_st, W.dw_self = (dw("long_term_attn.dw_conv.conv.weight"),
                                           dw("short_term_attn.dw_conv.conv.weight"),
                                           dw("self_attn.dw_conv.conv.weight"))
            W.Wp_ls = self._pl(torch.cat([g("long_term_attn.projection.weight"),
                                          g("short_term_attn.projection.weight")], dim=1))  # [512][2048]
            W.bp_ls = self._f(g("long_term_attn.projection.bias") + g("short_term_attn.projection.bias"))
            W.Wrel = self._pl(g("short_term_attn.relative_emb_k.weight").reshape(self.WIN, self.DATT))
            W.brel = self._f(g("short_term_attn.relative_emb_k.bias"))
            # The relative-position bias (attention.py:314: a 1x1 conv of the UNSCALED key = Q projection)
            # and the temporal-PE bias (transformer.py:1140-1172: (Q + cur_pe) . mem_pe[row]) are linear in
            # the layer input x = norm1(tgt): with the weight products formed once in fp64 they become two
            # more members of the grouped Q / V / U launch instead of two dependent launches after it.
            wq64, bq64 = qv_w[:self.DATT].double(), qv_b[:self.DATT].double()
            wrel64 = g("short_term_attn.relative_emb_k.weight").reshape(self.WIN, self.DATT).double()
            W.Wrel_x = self._pl((wrel64 @ wq64).float())                                   # [225][256]
            W.brel_x = self._f((wrel64 @ bq64 + g("short_term_attn.relative_emb_k.bias").double()).float())
            mem64, cur64 = sd["mem_pos_emb"].double(), sd["cur_pos_emb"][0].double()
            pe4_w, pe4_b = mem64 @ wq64, mem64 @ (bq64 + cur64)                            # [4][256], [4]
            W.pe_x = {}                                                                    # T -> (planes [T][256], bias [T])
            for T in range(1, self.cfg.FORMER_MEM_LEN + self.cfg.LATTER_MEM_LEN + 2):
                r = temporal_pe_rows(T)
                W.pe_x[T] = (self._pl(pe4_w[r].float()), self._f(pe4_b[r].float()))
            W.Wqk, W.bqk = self._pl(g("self_attn.linear_QK.weight")), self._f(g("self_attn.linear_QK.bias"))
            W.Wv12 = self._pl(torch.cat([g("self_attn.linear_V1.weight"), g("self_attn.linear_V2.weight")], 0))
            W.bv12 = self._f(torch.cat([g("self_attn.linear_V1.bias"), g("self_attn.linear_V2.bias")], 0))
            W.Wu12 = self._pl(torch.cat([g("self_attn.linear_U1.weight"), g("self_attn.linear_U2.weight")], 0))
            W.bu12 = self._f(torch.cat([g("self_attn.linear_U1.bias"), g("self_attn.linear_U2.bias")], 0))
            W.Wp_self, W.bp_self = (self._pl(g("self_attn.projection.weight")),
                                    self._f(g("self_attn.projection.bias")))
            # the weights of the projections that read the normalised rows, in MFMA-fragment order for
            self.lw.append(W)

    # ------------------------------------------------------------------ key splits
    @staticmethod
    def choose_splits(N: int, h: int, w: int, cap: int, clips: int = 1, cus: int = 256):
        """(ks_long, ks_win) of the paired long-term + windowed read launch: the pair that minimises the
        launch's makespan on `cus` CUs under a simple cost model measured on MI355X
        (profiles/r03f_kbench_read.json, k cycles per 64-key tile: long-term 9.5, windowed 14.6 -- relative-bias
        gathers and window arithmetic; ~14 per unit for the Q staging, the first tiles' scores, statistics and
        the flush; the measured sweep around the choice is profiles/r03y_split_sweep.txt).  Units are dispatched long-term first; with more units
        than CUs the surplus starts as the first units finish.  Several clips per launch share the CUs."""
        nq = (N + 63) // 64
        tv = (N + 63) // 64
        if clips > 1:
            # several clips per launch: two rounds of workgroups (longer units amortise the per-unit staging and flush);
            # measured at 480p K=4 (frames/s for long,win,self: 4 clips 3,1,4 479 / 4,2,4 476 / 7,2,6 461;
            # 8 clips 2,1,2 492 / 3,1,3 489 / 1,1,2 477) -- the model below does not cover the clip-major dispatch order
            total = max(3, min(32, 2 * cus // (nq * clips)))
            kw = max(1, min(8, int(round(total * 0.33 * min(1.0, 4.0 / max(cap, 1))))))
            return max(1, total - kw), kw
        long_tiles = cap * tv
        band = min(tv, ((min(h, 3 + 14) * w) + 63) // 64 + 1)       # key tiles visible to a 64-query tile (15x15 window)
        LONG, WIN, FIX = 9.5, 14.6, 14.0
        best = None
        cands = []
        for kl in range(1, 33):
            for kw in range(1, 9):
                if kl > long_tiles or kw > band:
                    continue
                cl = -(-long_tiles // kl) * LONG + FIX
                cw = -(-band // kw) * WIN + FIX
                units = [cl] * (nq * kl * clips) + [cw] * (nq * kw * clips)
                if len(units) <= cus:
                    span = max(cl, cw)
                elif len(units) > 4 * cus:
                    continue
                else:                                               # greedy list schedule in dispatch order
                    import heapq
                    free = [0.0] * cus
                    heapq.heapify(free)
                    span = 0.0
                    for c in units:
                        t0 = heapq.heappop(free)
                        heapq.heappush(free, t0 + c)
                        span = max(span, t0 + c)
                key = (round(span, 1), kl + kw)
                cands.append((span, kl, kw))
                if best is None or key < best[0]:
                    best = (key, kl, kw)
        if best is None:
            # beyond 4 rounds of units per CU whatever the split (N > 32768 tokens): no key split pays -- a query
            # tile's keys stay in one unit
            return 1, 1
        kl, kw = best[1], best[2]
        # One clip whose shortest launch fits the machine in one round: take the FEWEST units whose launch stays within 1.5 x
        # that shortest one.  The frame is bound by the CU-time its kernels hold, not by the length of this launch (what the
        # read leaves free the prefetched encoder pass uses): fewer, longer units carry fewer prologues, flushes and split
        # partials, and a windowed read in ONE split needs no partial or combine at all (rmem_read_args.gout).  Measured at
        # 480p K=4, same box, alternating (profiles/r06j_ks_sweep_fused.txt): (7, 2) 513.5 / 513.0 frames/s with a 110 us
        # launch; (5, 1) 521.5 / 522.4 with a 140 us launch; (6, 1) 509 / 512, (7, 1) 514 / 511, (4, 1) 517 (the model's
        # spans: 166 / 233 / 233 / 233 / 275 k cycles).  `RMEM_KS` overrides.
        if clips == 1 and nq * (kl + kw) <= cus and os.environ.get("RMEM_SPLITS_SHORTEST") != "1":
            lim = 1.5 * best[0][0]
            ok = [(a + b, sp, a, b) for sp, a, b in cands if sp <= lim and nq * (a + b) <= cus]
            if ok:
                _, _, kl, kw = min(ok)
        # More units than CUs: the windowed units (last in dispatch order) queue on the few CUs per XCD the long-term
        # units leave free and end up as the makespan (720p K=8: 8 units of up to 22 tiles on 2 CUs per XCD, at the
        # throttled clock); halves of them pack better.  Measured (profiles/r03q_split_sweep_720p.txt): (4, 1) 743 us,
        # (4, 2) 683, (4, 3) 675, (4, 4) 676 per launch; 193.8 -> 196.8 frames/s.
        if nq * (kl + kw) * clips > cus and kw == 1 and band >= 8:
            kw = 2
        return kl, kw

    @staticmethod
    def window_unit_tiles(N: int, h: int, w: int, kw: int):
        """Key tiles of every windowed unit (query tile x split), computed as read64.hip does (15 x 15 window)."""
        out = []
        for qt in range((N + 63) // 64):
            q_lo, q_hi = qt * 64, min(qt * 64 + 63, N - 1)
            y_lo, y_hi = max(q_lo // w - 7, 0), min(q_hi // w + 7, h - 1)
            k_lo, k_hi = (y_lo * w) // 64, ((y_hi + 1) * w + 63) // 64
            per = -(-(k_hi - k_lo) // kw)
            for z in range(kw):
                out.append(max(0, min(k_lo + z * per + per, k_hi) - (k_lo + z * per)))
        return out

    @staticmethod
    def choose_uneven(N: int, h: int, w: int, T: int, kl: int, kw: int, cus: int = 256):
        """Uneven long-term splits for the paired read of a bank of T slots (rmem_read_args.nfull / pf), or None when the
        even pair (kl, kw) is as good.  With an even split the launch lasts as long as its longest unit while the
        windowed units (4-8 key tiles against 16) leave their CUs early and a few CUs get no unit at all (480p K=4: 189 +
        54 units on 256 CUs, 15 % of the CU-time idle).  Here the first `nfull` splits of a query tile hold `pf` key tiles
        each and `ns` short splits share the rest; rmem_attn_read2 queues long pieces, windowed units, short pieces in
        that order (longest first), so the short pieces land on the CUs the windowed units free.  Cost model in k cycles,
        from the per-unit stamps of profiles/r03f_kbench_read.json: 9.0 per long-term tile + 13.3 per unit, 13.3 per
        windowed tile + 9.3 per unit (the windowed units' tile counts are exact: window_unit_tiles), + 3 per extra partial
        for the combine's traffic; the candidate must beat the even pair by 4 %."""
        import heapq
        nq = tv = (N + 63) // 64
        long_tiles = T * tv
        LONG, FIXL, WIN, FIXW, PART = 9.0, 13.3, 13.3, 9.3, 3.0
        win_units = sorted((t * WIN + FIXW for t in DeAOTLSTT.window_unit_tiles(N, h, w, kw) if t > 0), reverse=True)

        def span(units):                                   # list scheduling in queue order
            if len(units) <= cus:
                return max(units)
            free = [0.0] * cus
            heapq.heapify(free)
            m = 0.0
            for c in units:
                t0 = heapq.heappop(free)
                heapq.heappush(free, t0 + c)
                m = max(m, t0 + c)
            return m

        per = -(-long_tiles // kl)
        even = span([min(per, long_tiles - z * per) * LONG + FIXL for z in range(kl) for _ in range(nq)] + win_units)
        best = None
        for nfull in range(1, 17):
            for pf in range(2, long_tiles):
                rest = long_tiles - nfull * pf
                if rest <= 0:
                    break
                for ns in range(1, 5):
                    if nfull + ns > 16 or nq * (nfull + ns + kw) > 3 * cus:
                        continue
                    ps = -(-rest // ns)
                    if ps >= pf or (ns - 1) * ps >= rest:      # short pieces are the short ones; none of them empty
                        continue
                    short = [min(ps, rest - z * ps) * LONG + FIXL for z in range(ns) for _ in range(nq)]
                    units = [pf * LONG + FIXL] * (nq * nfull) + win_units + short
                    cost = span(units) + PART * (nfull + ns - kl)
                    if best is None or cost < best[0]:
                        best = (cost, nfull + ns, nfull, pf)
        if best is None or best[0] > 0.96 * even:
            return None
        return best[1], best[2], best[3]

    # ------------------------------------------------------------------ buffers
    def _alloc(self):
        N, Np, dev = self.N, self.Npad, self.dev
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)
        self.tgt, self.tgt_id = z(N, 256), z(N, 256)
        self._tg, self._tgi = self.tgt, self.tgt_id
        self.sched = None if os.environ.get("RMEM_NO_PULL") == "1" else z(2, dt=torch.int32)   # rmem_read_args.sched
        self.x_pl = Planes.empty((Np, 256), dev)
        self.z_pl = [Planes.empty((Np, 256), dev) for _ in range(self.L)]
        self.idemb_pl = Planes.empty((Np, 256), dev)
        self.Qpe = Planes.empty((Np, 128), dev)
        self.bankK = [Planes.empty((self.S, Np, 128), dev) for _ in range(self.L)]
        self.bankV = [Planes.empty((self.S, Np // 16, 1024, 16), dev) for _ in range(self.L)]   # blocked-16
        self.Ucat0 = z(N, 1024)
        self.Ucat0[:, 512:] = 1.0                                 # layer 0: gate of the ID half is 1
        self.Ucat = z(N, 1024)
        self.bias_pe = z(N, self.Tmax)
        self._layer = 0
        # Key splits of the fused reads (csrc/read64.hip).  A unit = (64-query tile, key split) is one 8-wave
        # workgroup that owns a CU; the long-term and the windowed read of a layer share ONE launch, the
        # self read gets a launch of its own.
        # CUs of the device the library will launch on (its own pull-versus-plain decision reads the same attribute,
        # read64.hip:device_cus): a partitioned or smaller part gets splits that match
        cus = int(torch.cuda.get_device_properties(dev).multi_processor_count) if dev.type == "cuda" else 256
        self.cus = cus = cus if cus > 0 else 256
        self.ks_long, self.ks_win = self.choose_splits(N, self.h, self.w, self.cap, self.clips_per_launch, cus)
        nq = (N + 63) // 64
        tv = (N + 63) // 64
        budget = cus if self.clips_per_launch == 1 else max(1, 2 * cus // self.clips_per_launch)
        # self read (T = 1, 27 tiles at 480p): 9 splits are faster isolated (33.8 + 13.1 us read + combine against
        # 38.2 + 10.1 with 6) but not in the frame: the rate follows the CU-time a launch holds, not its isolated latency
        # (DESIGN.md section 9).  In-frame A/B on one box, three alternating repeats (profiles/r05n_split_sweep.txt):
        # 6 splits 521.6 / 523.1 / 523.2 frames/s, 4 splits 528.0 / 528.6 / 529.0, 3 splits 526.2 / 527.3 / 527.1, 1: 504.
        self.ks_self = max(1, min(budget // nq, tv, 4))
        # Uneven long-term splits of the paired read, per bank depth T: OFF by default (RMEM_UNEVEN=1 turns the chooser
        # on).  Measured at 480p K=4 (profiles/r04n_split_sweep.txt): (7 x 15 + 3) 110.5 us, (7 x 14 + 2 x 5) 123.0 us
        # against 101.6 us for the even (7, 2) -- with more units than CUs the launch goes through the unit queue, and the
        # extra partials and per-unit fixed costs outweigh the idle CU-time the model promised to recover.
        self.uneven: Dict[int, tuple] = {}
        if self.clips_per_launch == 1 and os.environ.get("RMEM_UNEVEN", "0") == "1":
            for T in range(1, self.Tmax + 1):
                u = self.choose_uneven(N, self.h, self.w, T, min(self.ks_long, T * tv), self.ks_win, cus)
                if u is not None:
                    self.uneven[T] = u
        if os.environ.get("RMEM_KS"):                             # tuning override: "long,win,self[,nfull,pf]" (uneven: every T)
            v = [int(x) for x in os.environ["RMEM_KS"].split(",")]
            self.ks_long, self.ks_win, self.ks_self = v[:3]
            self.uneven = {T: (v[0], v[3], v[4]) for T in range(1, self.Tmax + 1) if v[3] * v[4] < T * tv} if len(v) >= 5 else {}
        self.ksplits_max = max([self.ks_long, self.ks_win, self.ks_self] + [u[0] for u in self.uneven.values()])
        self.ws_main = _AttnWS(self.Tmax, N, Np, max([self.ks_long, self.ks_self] + [u[0] for u in self.uneven.values()]), dev)
        self.ws_side = _AttnWS(1, N, Np, self.ks_win, dev)
        self.branch_order = os.environ.get("RMEM_BRANCH_ORDER", "serial")   # serial (paired launches) | serial_unpaired
        # Layer 0's long-term read needs the current frame's Q and the bank only -- not the previous frame's label -- so on
        # a front / rest split it can run with the front part (hoisted beside the previous frame's decoder,
        # DeAOTEngine._try_hoist); the windowed read and the combine stay in `rest`.  Same kernels' arithmetic, separate
        # launches for that layer (bit-identical: the split units are those of the paired launch).
        self.early_long_read = os.environ.get("RMEM_EARLY_LONG_READ", "0") == "1"
        if self.early_long_read and os.environ.get("RMEM_UNEVEN") == "1":
            # (the early launch goes through rmem_attn_read, which honours even key splits only; the sampled-read timing of
            # bench.py would also time a schedule that is not the one that runs)
            raise hip.RmemError("RMEM_EARLY_LONG_READ=1 cannot be combined with RMEM_UNEVEN=1")
        self.Ylt = Planes.empty((Np, 1024), dev)
        self.Yst = Planes.empty((Np, 1024), dev)
        # split-K of the projection GEMMs.  Under the streaming kernel two splits are one round of 216 items (19.1 / 12.9 us
        # for the two projections against 22.3 / 15.6 with four, profiles/r04y_kbench.json) and half the partials for
        # the LayerNorm that folds them; the tile-per-workgroup kernels (recorded launches, RMEM_LINEAR=tiles) are
        # fastest with four (21.2 against 27.4 us)
        self.KS = 2 if (self.clips_per_launch == 1 and not self._force_tiles) else 4
        if os.environ.get("RMEM_PROJ_KS"):                       # tuning override
            self.KS = int(os.environ["RMEM_PROJ_KS"])
        self.parts = z(self.KS, N, 512)
        # relative bias of the windowed read, stored by anti-diagonals: element (q, o) at 225*(q+o) + o
        # = q*ldr + o*rcs with ldr = 225, rcs = 226 (rmem_read_args.rcs: coalesced gathers)
        self.ldr, self.rcs = 225, 226
        self.R = z(N * 225 + 225 * 226)
        self.s_pl = Planes.empty((Np, 512), dev)
        self.selfQK = Planes.empty((1, Np, 128), dev)
        self.selfV = Planes.empty((1, Np // 16, 1024, 16), dev)
        self.Uself = z(N, 1024)
        self.out = z(N, 512)
        self.gn_ws = z(4 * ((N + 63) // 64), dt=torch.float64)
        self.mass = z(N, self.Tmax)
        self.w_out = z(self.Tmax)
        self.maps = z(32, dt=torch.int32)                         # [0:16] bank map, [16] short slot
        self.bank_state = z(C.sizeof(hip.BankState) // 4, dt=torch.int32)       # rmem_bank_state (device-side eviction)
        self.policy_result = torch.zeros(4, dtype=torch.int32).pin_memory() if self.dev.type == "cuda" else \
            torch.zeros(4, dtype=torch.int32)
        self.fg = z(N)
        self.k_slot_stride = Np * 128
        self.v_slot_stride = 1024 * Np

    # ------------------------------------------------------------------ measurement
    def enable_kernel_timing(self, on: bool):
        """HIP-event pairs around the dominant kernel (the fused long-term + windowed read of a
        layer, read64x2_kernel) on the launch stream."""
        self._timing = bool(on)
        if on:
            self._events = []

    def launch_read2_layer0(self):
        """The fused long-term + windowed read of layer 0 for the frame _prepare() set up, launched on its own between
        two HIP events (appended to the list roofline_report() reads).  DeAOTEngine._graphed_frame uses it on sampled
        frames between the replayed `front` graph and the `tail` graph (= `rest` without this launch): the kernel is
        then timed inside a REPLAYED frame, beside the prefetched encoder pass, like every other frame of the timed
        region (torch on ROCm refuses event-record nodes inside a hipGraph: profiles/r04a_graph_event_probe.json)."""
        T, cur = self._T, self.cur
        self._layer = 0
        A = self._read_args(self.ws_main, 0, T, self.bankK[0], self.bankV[0], self.maps.data_ptr(), self.Qpe,
                            self.bias_pe, self.Ucat0, True, self.ks_long, uneven=True)
        B = self._read_args(self.ws_side, 1, 1, self.bankK[0], self.bankV[0], self.maps.data_ptr() + 16 * 4,
                            self.bankK[0][cur], None, self.Ucat0, False, self.ks_win)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.check(hip.load().rmem_attn_read2(C.byref(A[0]), C.byref(B[0]), hip.stream_ptr()), "rmem_attn_read2")
        e1.record()
        self._events.append((e0, e1, T))

    def read_flops(self, T: int) -> float:
        """Algorithmic FLOPs of one read2 launch: Q.K^T and P.V of the
        long-term read over T*N keys plus the 225-key windowed read, unpadded, one product each."""
        return 2.0 * self.N * (T * self.N + self.WIN) * (1024 + 128)

    def roofline_report(self, mfma_peak_tflops: float):
        if not self._events:
            return None
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b, _ in self._events]
        T = self._events[0][2]
        flops = self.read_flops(T)
        mean_ms = sum(ms) / len(ms)
        ach = flops / (mean_ms * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": f"read64x2_kernel / read64x2_pull_kernel (fused long-term T={T} + windowed memory read: Q.K^T, softmax, P.V)",
                "achieved": ach, "peak": mfma_peak_tflops, "unit": "TFLOP/s", "frac": ach / mfma_peak_tflops,
                "traffic": None, "launches": len(ms), "mean_us": 1e3 * mean_ms,
                "algorithmic_flops_per_launch": flops}

    def time_read_isolated(self, iters: int = 20) -> float:
        """Mean duration (us) of the read2 launch of the current bank state with nothing else on
        the GPU (back-to-back launches between two HIP events)."""
        T = len(self.bank)
        self._layer = 0
        A = self._read_args(self.ws_main, 0, T, self.bankK[0], self.bankV[0], self.maps.data_ptr(), self.Qpe,
                            self.bias_pe, self.Ucat0, True, self.ks_long, uneven=True)
        B = self._read_args(self.ws_side, 1, 1, self.bankK[0], self.bankV[0], self.maps.data_ptr() + 64,
                            Planes(self.bankK[0].hi[self.cur], self.bankK[0].lo[self.cur]), None, self.Ucat0, False,
                            self.ks_win)
        lib, st = hip.load(), hip.stream_ptr()
        torch.cuda.synchronize()
        for _ in range(3):
            hip.check(lib.rmem_attn_read2(C.byref(A[0]), C.byref(B[0]), st), "rmem_attn_read2")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            hip.check(lib.rmem_attn_read2(C.byref(A[0]), C.byref(B[0]), st), "rmem_attn_read2")
        e1.record()
        e1.synchronize()
        return 1e3 * e0.elapsed_time(e1) / iters

    def clear_memory(self):                                        # transformer.py:1000-1007
        self.bank: List[int] = []          # logical -> physical slot
        self.short: Optional[int] = None
        self.cur: int = 0
        self.mass_T = 0
        self.ema: Dict[int, float] = {}
        self.visits: Dict[int, int] = {}
        self._pending = None               # device-side eviction whose result the host has not read yet
        self._policy_seq = getattr(self, "_policy_seq", 0)

    @property
    def _dev_policy(self) -> bool:
        """Eviction rule on the device (rmem_bank_policy_step) -- the default for one clip per launch, and for the clips of
        rmem_amd.batched (BatchedLSTT switches it on: per clip the same tiny launches, no device-to-host copy on the
        critical path); a DeAOTLSTT built with clips_per_launch > 1 on its own keeps the rule on the host."""
        return self.device_policy

    def resolve_policy(self, block: bool = True) -> bool:
        """Bring the host's view of the bank (self.bank, the engine's long_memories_indexes) up to date with the
        last device-side eviction.  block=False: only if its result has already arrived."""
        p = self._pending
        if p is None:
            return True
        if not block and not p["event"].query():
            return False
        hip.wait_event(p["event"])        # (the host is up to `gap` frames ahead: the one long wait of the frame loop; sleeps, does not spin)
        res = self.policy_result
        if int(res[0]) != p["seq"]:
            raise hip.RmemError(f"eviction result out of sequence ({int(res[0])} != {p['seq']})")
        drop = int(res[1])
        self._pending = None
        self.last_policy = dict(drop=drop)
        left = int(res[2])
        if p["expect_drop"]:
            if drop < 0 or left != len(self.bank) - 1:
                raise hip.RmemError("device-side eviction disagrees with the host's bank length")
            del self.bank[drop]
            p["indexes"].remove(p["indexes"][drop])
        elif drop >= 0 or left != len(self.bank):
            # the host expected no eviction (bank below its cap): the device must not have dropped a slot either
            raise hip.RmemError(f"device-side eviction dropped position {drop} ({left} slots left) although the "
                                f"host's bank holds {len(self.bank)} <= cap slots")
        return True

    # ------------------------------------------------------------------ helpers
    def _sync_maps(self):
        m = torch.zeros(32, dtype=torch.int32)
        for i, s in enumerate(self.bank):
            m[i] = s
        m[16] = self.short if self.short is not None else 0
        self.maps.copy_(m, non_blocking=False)

    def _tile(self, tiles: int) -> int:
        """rmem_linear_args.tile of a projection: 0 = the streaming kernel (csrc/linear.hip: one persistent workgroup per
        CU, operands by LDS-DMA) for launches issued directly; recorded launches (several clips per launch,
        rmem_amd.batched) keep the tile-per-workgroup kernel `tiles`.  Bit-identical either way
        (tests/test_hip_ops.py::test_linear_stream_equals_tile_kernels); RMEM_LINEAR=tiles forces the tile kernels."""
        return tiles if (self._batched or self._force_tiles) else 0

    def _free_slot(self) -> int:
        used = set(self.bank)
        if self.short is not None:
            used.add(self.short)
        for s in range(self.S):
            if s not in used:
                return s
        raise hip.RmemError("no free bank slot (internal error)")

    def _ln(self, x, gb, out: Planes, ldo, col_off=0, parts_col=None):
        """LayerNorm -> planes; parts_col = column offset into the split-K partials of the
        preceding projection GEMM, which are first summed into x (fixed order)."""
        np_, pp = (self.KS, self.parts.data_ptr() + parts_col * 4) if parts_col is not None else (0, None)
        rc = hip.load().rmem_layernorm_red(
            x.data_ptr(), 256, pp, np_, self.N * 512, 512, gb[0].data_ptr(), gb[1].data_ptr(), self.N, 256, 1e-5,
            out.hi.data_ptr() + col_off * 2, out.lo.data_ptr() + col_off * 2, ldo, None, 0, hip.stream_ptr())
        hip.check(rc, "rmem_layernorm_red")

    def _ln2(self, gb0, out0: Planes, ldo0, off0, gb1, out1: Planes, ldo1, off1, parts: bool):
        """LayerNorm of tgt (-> out0) and of tgt_id (-> out1) in one launch; parts: first fold the
        split-K partials of the preceding projection (columns 0.. / 256..) into the two streams."""
        np_ = self.KS if parts else 0
        pp0 = self.parts.data_ptr() if parts else None
        pp1 = self.parts.data_ptr() + 256 * 4 if parts else None
        with self._ev("layernorm_red2_kernel", nbytes=self.N * 512 * 4.0 * (np_ + 3)):
            rc = self._ln2_launch(np_, pp0, pp1, gb0, out0, ldo0, off0, gb1, out1, ldo1, off1)
        hip.check(rc, "rmem_layernorm_red2")

    def _ln2_launch(self, np_, pp0, pp1, gb0, out0, ldo0, off0, gb1, out1, ldo1, off1):
        return hip.load().rmem_layernorm_red2(
            self._tg.data_ptr(), self._tgi.data_ptr(), 256, pp0, pp1, np_, self.N * 512, 512,
            gb0[0].data_ptr(), gb0[1].data_ptr(), gb1[0].data_ptr(), gb1[1].data_ptr(), self.N, 256, 1e-5,
            out0.hi.data_ptr() + off0 * 2, out0.lo.data_ptr() + off0 * 2, ldo0,
            out1.hi.data_ptr() + off1 * 2, out1.lo.data_ptr() + off1 * 2, ldo1, hip.stream_ptr())

    def _ev(self, name: str, flops: float = 0.0, nbytes: float = 0.0, overhead: float = 0.0):
        """Context manager: HIP events around one launch of the memory path while bench.py samples a frame
        (self._kev is a list); a no-op otherwise.  flops / nbytes = the launch's ALGORITHMIC work as SURVEY section 8d counts
        it (unpadded, one product per multiply-add, every operand read once and every final output written once; the bank
        and Q at 2 bytes per element, fp32 outputs; projections with fp32-equivalent operands).  overhead = bytes the
        design moves on top of that and no algorithm needs: the split partials of the key-split reads (written by the read,
        re-read by the combine) and of the split-K projections."""
        import contextlib
        if self._kev is None:
            return contextlib.nullcontext()

        @contextlib.contextmanager
        def cm():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            yield
            e1.record()
            self._kev.append((name, e0, e1, float(flops), float(nbytes), float(overhead)))
        return cm()

    def kernel_report(self, mfma_peak_tflops: float, hbm_peak_gbs: float = 8000.0):
        """Per kernel class of the sampled frames: launches per frame, mean duration, time per frame, algorithmic work and
        the fraction of the roofline that bounds the class (MFMA for the contractions, HBM for the element-wise ones)."""
        if not self._kev_store or not self._kev_frames:
            return []
        torch.cuda.synchronize()
        frames = self._kev_frames
        acc: Dict[str, list] = {}
        for name, e0, e1, fl, by, ov in self._kev_store:
            a = acc.setdefault(name, [0, 0.0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += e0.elapsed_time(e1) * 1e3
            a[2] += fl
            a[3] += by
            a[4] += ov
        out = []
        for name, (n, us, fl, by, ov) in acc.items():
            ent = {"kernel": name, "launches_per_frame": n / frames, "mean_us": us / n, "us_per_frame": us / frames,
                   "algorithmic_mb_per_launch": by / n / 1e6, "overhead_mb_per_launch": ov / n / 1e6}
            if fl > 0:
                ach = fl / (us * 1e-6) / 1e12
                ent.update(bound="mfma", algorithmic_gflop_per_launch=fl / n / 1e9, achieved=ach, peak=mfma_peak_tflops,
                           unit="TFLOP/s", frac=ach / mfma_peak_tflops)
            else:
                # bandwidth-bound classes: `achieved` counts the ALGORITHMIC bytes only (a kernel that moves nothing but
                # split partials -- the combine -- achieves 0); what it really moves per second is `moved_gbs`
                ach = by / (us * 1e-6) / 1e9
                ent.update(bound="hbm", achieved=ach, peak=hbm_peak_gbs, unit="GB/s", frac=ach / hbm_peak_gbs,
                           moved_gbs=(by + ov) / (us * 1e-6) / 1e9)
            out.append(ent)
        out.sort(key=lambda e: -e["us_per_frame"])
        return out

    def _read_args(self, ws: "_AttnWS", mode: int, T: int, kpl: Planes, vpl: Planes, slot_map_ptr,
                   qpl: Planes, bias, U, want_mass: bool, ksplits: int, uneven: bool = False):
        """Argument blocks (fused read, combine) of one read into workspace `ws`.  A read in ONE split that records no
        attention mass writes the gated aggregate itself (rmem_read_args.gate / gout): its combine block is unused."""
        Np = self.Npad
        ra = hip.ReadArgs()
        ra.mode, ra.qh, ra.ql = mode, qpl.hi.data_ptr(), qpl.lo.data_ptr()
        ra.kh, ra.kl, ra.k_slot_stride = kpl.hi.data_ptr(), kpl.lo.data_ptr(), self.k_slot_stride
        ra.vh, ra.vl, ra.v_slot_stride = vpl.hi.data_ptr(), vpl.lo.data_ptr(), self.v_slot_stride
        ra.slot_map, ra.T, ra.N, ra.Npad, ra.ncols = slot_map_ptr, T, self.N, Np, 1024
        ra.scale = self.scale
        ra.bias = bias.data_ptr() if bias is not None else None
        ra.R, ra.ldr, ra.h, ra.w = (self.R.data_ptr() if mode == 1 else None), self.ldr, self.h, self.w
        ra.rcs = self.rcs
        tiles = T * ((self.N + 63) // 64)
        ks = max(1, min(ksplits, tiles))
        if uneven and mode == 0 and T in self.uneven and not self._batched:
            ks, ra.nfull, ra.pf = self.uneven[T]
        ra.ksplits = ks
        ra.part, ra.ml = ws.part.data_ptr(), ws.ml.data_ptr()
        ra.lslot = ws.lslot.data_ptr() if want_mass else None
        # unit queue of the paired read (used by the library only when a launch holds more units than the device has
        # CUs: 720p K=8, several clips per launch); RMEM_NO_PULL=1 leaves the surplus to the hardware's dispatch order
        ra.sched = self.sched.data_ptr() if (mode == 0 and self.sched is not None) else None
        if ks == 1 and not want_mass and self.fuse_gate:
            ra.gate, ra.ldgate, ra.gout, ra.ldgout = U.data_ptr(), 1024, ws.G.data_ptr(), 1024
        ca = hip.ReadCombineArgs()
        ca.T, ca.N, ca.Npad, ca.ncols, ca.ksplits = T, self.N, Np, 1024, ks
        ca.part, ca.ml, ca.lslot = ws.part.data_ptr(), ws.ml.data_ptr(), (ws.lslot.data_ptr() if want_mass else None)
        ca.U, ca.ldu, ca.G, ca.ldg = U.data_ptr(), 1024, ws.G.data_ptr(), 1024
        ca.mass = self.mass.data_ptr() if want_mass else None
        return ra, ca

    def _read(self, A):
        """One fused read + its combine -> ws.G."""
        lib, st = hip.load(), hip.stream_ptr()
        N, ks = self.N, A[0].ksplits
        with self._ev("read64_kernel (self read: T=1, Q=K)", flops=2.0 * N * A[0].T * N * (1024 + 128),
                      nbytes=N * (2.0 * (A[0].T * 1152 + 128) + 4.0 * 1024), overhead=ks * N * 4096.0):
            hip.check(lib.rmem_attn_read(C.byref(A[0]), st), "rmem_attn_read")
        if A[0].gout:            # (one split: the read gated its own output)
            return
        with self._ev("read_combine_kernel", nbytes=0.0, overhead=(ks + 2) * N * 4096.0):
            hip.check(lib.rmem_attn_read_combine(C.byref(A[1]), st), "rmem_attn_read_combine")

    def _read_pair(self, A, B, long_done: bool = False):
        """The long-term (A) and windowed (B) reads of a layer: ONE read launch, ONE combine launch.  long_done: the
        long-term read was launched with the front part (early_long_read) -- only the windowed read is left."""
        lib, st = hip.load(), hip.stream_ptr()
        def combine():
            """Merges the key splits of the reads that have any (a read in one split gated its own output)."""
            fa, fb = bool(A[0].gout), bool(B[0].gout)
            if fa and fb:
                return
            if fa or fb:
                X = B if fa else A
                with self._ev("read_combine_kernel", nbytes=0.0, overhead=(X[0].ksplits + 2) * self.N * 4096.0):
                    hip.check(lib.rmem_attn_read_combine(C.byref(X[1]), st), "rmem_attn_read_combine")
                return
            with self._ev("read_combine2_kernel", nbytes=0.0, overhead=(A[0].ksplits + B[0].ksplits + 4) * self.N * 4096.0):
                hip.check(lib.rmem_attn_read_combine2(C.byref(A[1]), C.byref(B[1]), st), "rmem_attn_read_combine2")

        if long_done:
            hip.check(lib.rmem_attn_read(C.byref(B[0]), st), "rmem_attn_read")
            combine()
            return
        if self._timing:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if self._skip_read2:               # part "tail": this launch is issued separately (launch_read2_layer0)
            self._skip_read2 = False
        else:
            # algorithmic bytes = SURVEY 8d's "long-term kernel I/O" (68.1 MB per frame / 3 at 480p K=4: bank and Q at 2
            # bytes, fp32 output); the windowed read's operands (one more slot, the relative bias, its output) are NOT added
            with self._ev("read64x2_kernel (long-term + windowed read)", flops=self.read_flops(A[0].T),
                          nbytes=self.N * (2.0 * (A[0].T * 1152 + 128) + 4.0 * 1024),
                          overhead=((0 if A[0].gout else A[0].ksplits) + (0 if B[0].gout else B[0].ksplits)) * self.N * 4096.0):
                hip.check(lib.rmem_attn_read2(C.byref(A[0]), C.byref(B[0]), st), "rmem_attn_read2")
        if self._timing:
            e1.record()
            self._events.append((e0, e1, A[0].T))
        combine()

    def _dwconv(self, ws: "_AttnWS", wt, out: Planes):
        with self._ev("dwconv5x5_split_kernel", nbytes=self.N * 1024 * 8.0):
            rc = hip.load().rmem_dwconv5x5_split(ws.G.data_ptr(), 1024, wt.data_ptr(), self.h, self.w, 1024,
                                                out.hi.data_ptr(), out.lo.data_ptr(), 1024, hip.stream_ptr())
        hip.check(rc, "rmem_dwconv5x5_split")

    def _idv(self, l: int, slot: int, launch: bool = True):
        """ID_V = silu(linear_ID_V([z | id_emb])) -> columns 512.. of the V planes of `slot`
        (fuse_key_value_id, transformer.py:1238-1244)."""
        W, N = self.lw[l], self.N
        dst = self.bankV[l][slot]
        if l == 0:
            return hip.linear(self.idemb_pl, W.Widv, N, 512, 256, ldx=256, ldy=256, bias=W.bidv, act=1,
                              pa=dst, ldpa=1024, pa_blocked=True, pa_off=512 * 16, nsplit=self.nsplit, tile=self._tile(64),
                              launch=launch)
        return hip.linear(self.z_pl[l], W.Widv, N, 512, 512, ldx=256, ldy=512, x2=self.idemb_pl, ldx2=256,
                          kx_split=256, bias=W.bidv, act=1, pa=dst, ldpa=1024, pa_blocked=True, pa_off=512 * 16,
                          nsplit=self.nsplit, tile=self._tile(64), launch=launch)

    # ------------------------------------------------------------------ ID assignment
    def assign_identity(self, label_u8: torch.Tensor, ignore: bool = True):
        """label [H][W] uint8 on device -> id_emb planes (aot_engine.py:208-232, deaot.py:65-69).
        ignore=True: label 255 selects the ignore channel (update_short_term_memory, :330-336);
        ignore=False: label 255 contributes nothing (add_reference_frame, :304 -> :209-213)."""
        H, Wd = label_u8.shape
        rc = hip.load().rmem_id_assign(
            label_u8.data_ptr(), H, Wd, self.id_wt.data_ptr(), self.id_bias.data_ptr(), self.id_ncls,
            self.id_ksize, self.id_stride, self.id_pad, self.h, self.w, 256, self.id_gamma.data_ptr(),
            self.id_beta.data_ptr(), 1e-5, self.idemb_pl.hi.data_ptr(), self.idemb_pl.lo.data_ptr(), 256,
            None, 0, int(bool(ignore)), hip.stream_ptr())
        hip.check(rc, "rmem_id_assign")

    # ------------------------------------------------------------------ forward
    def forward(self, emb_nc: torch.Tensor, ref_frame: bool = False) -> torch.Tensor:
        """DualBranchGPM.forward (transformer.py:765-824).  emb_nc: [N,256] fp32 on device.
        ref_frame=True is the curr_id_emb branch (:1125-1135): the frame attends itself
        and becomes bank slot 0; ``assign_identity`` must have been called before."""
        self.tgt.copy_(emb_nc)
        self._prepare(ref_frame)
        self._forward_device(ref_frame)
        return self._finish(ref_frame)

    def _prepare(self, ref_frame: bool):
        """Host part of a forward pass: pick the slot the frame is written to and publish the
        logical->physical slot map (one small H2D copy).  Nothing here is baked into a graph."""
        self.resolve_policy(block=False)
        self.cur = self._free_slot()
        if ref_frame:
            bank_map, short = [self.cur], self.cur
        else:
            bank_map, short = self.bank, self.short
        # (a device-side eviction the host has not read yet: the bank already holds one slot less)
        self._T = len(bank_map) - (1 if (not ref_frame and self._pending is not None and self._pending["expect_drop"]) else 0)
        # [0:16] bank map, [16] short-term slot of this frame, [17] short-term slot of the NEXT frame
        # (= this frame's slot): read by the next frame's hoisted front part, see _forward_device
        vals = list(bank_map)[:16] + [0] * (16 - len(bank_map)) + [short, self.cur]
        self._map_vals = vals
        if self._dev_policy:               # the bank map lives on the device: only the short-term slots are published
            if not self._batched:          # (batched: one upload of [short, cur] for all clips, rmem_amd.batched)
                hip.set_ints(self.maps, [short, self.cur], offset=16, count=2)
            if ref_frame:
                self._pending = None
                hip.check(hip.load().rmem_bank_reset(self.maps.data_ptr(), self.bank_state.data_ptr(), self.cur,
                                                     int(getattr(self, "ref_frame_index", 0)), hip.stream_ptr()), "rmem_bank_reset")
        elif not self._batched:            # batched: one upload for all clips (rmem_amd.batched)
            hip.set_ints(self.maps, vals)

    def next_free_slot(self) -> int:
        """The slot _prepare() will pick for the next frame if this frame's update does not touch
        the long-term bank (short := cur, bank unchanged)."""
        used = set(self.bank) | {self.cur}
        return next(s for s in range(self.S) if s not in used)

    def _finish(self, ref_frame: bool):
        self.mass_T = self._T
        if ref_frame:                      # init_memory (transformer.py:993-998)
            self.bank, self.short = [self.cur], self.cur
            self.ema, self.visits = {}, {}
        return self.out

    def graph_key(self):
        """Everything a captured forward depends on besides device memory contents."""
        return (self._T, self.cur)

    def buffer_signature(self):
        """Addresses of every device buffer a captured graph of this LSTT bakes in (unit-queue counters, slot maps, residual
        streams, planes, banks, read workspaces, ...).  The buffers are allocated once per LSTT, so the signature is constant
        for its lifetime; the engine records it with its graphs and compares at every clip start (DeAOTEngine.restart_engine)
        -- a graph must never be replayed against a pointer that has since been reallocated (advisor, round 5: the 720p
        second-clip fault was bisected to a memset node, but a stale pointer would have shown the same symptoms)."""
        ptrs = []
        for name in sorted(self.__dict__):
            if name.startswith("_"):           # (per-frame aliases and inputs: the encoder feature copy, the current stream pair)
                continue
            v = self.__dict__[name]
            for t in (v if isinstance(v, (list, tuple)) else [v]):
                if torch.is_tensor(t):
                    ptrs.append(t.data_ptr())
                elif isinstance(t, Planes):
                    ptrs.extend((t.hi.data_ptr(), t.lo.data_ptr()))
                elif isinstance(t, _AttnWS):
                    ptrs.extend(x.data_ptr() for x in t.__dict__.values() if torch.is_tensor(x))
        return hash(tuple(ptrs))

    def graph_variants(self):
        """Host states (attribute dicts) that differ only in graph_key() for the current T:
        all of them are captured together the first time one is needed."""
        return [{"cur": c} for c in range(self.S)]

    def _forward_device(self, ref_frame: bool = False, part: str = "all", src_cn: Optional[torch.Tensor] = None):
        """Device part of a forward pass (reads self.tgt, writes self.out): kernel launches and
        device-side memsets only -- hipGraph-capturable; depends on the host only through
        graph_key() = (T, cur).

        part = "front" / "rest" split the pass where the previous frame's label first matters: the
        front (layer 0: norm1, Q / V / U projections, relative-bias GEMM, temporal-PE bias) needs
        only the frame's encoder features, so the engine can issue it beside the previous frame's
        decoder (DeAOTEngine._try_hoist); "rest" starts at the fused memory reads, which read the
        ID values written by the previous frame's memory update and the slot maps."""
        N, Np, ns = self.N, self.Npad, self.nsplit
        lib, st = hip.load(), hip.stream_ptr()
        # "tail" = "rest" without its first launch, the fused read of layer 0 (launch_read2_layer0 issues that one
        # between HIP events on sampled frames of bench.py)
        do_front, do_rest = part not in ("rest", "tail"), part != "front"
        self._split_parts = part in ("front", "rest")
        self._skip_read2 = part == "tail"
        if part != "all" and (ref_frame or self.branch_order != "serial"):
            raise hip.RmemError("front/rest split needs a propagation frame and the paired schedule")
        cur, T = self.cur, self._T
        self._tg, self._tgi = self.tgt, self.tgt_id        # (the front part folds nothing: every part starts on the first pair)
        self._gn_fold = False
        map_bank = self.maps.data_ptr()
        map_short = self.maps.data_ptr() + 16 * 4
        rows = (C.c_int32 * 16)(*(temporal_pe_rows(T) + [0] * (16 - T)))
        # src_cn: the encoder's last feature map [256][h*w] (channel-major, contiguous): layer 0's norm1 then reads it
        # directly (rmem_layernorm_cn: transpose + residual stream + zeroed ID stream + planes in ONE launch) instead of
        # the caller's transposing copy into self.tgt, the fill of self.tgt_id and the LayerNorm launch
        self._src_cn = src_cn if (src_cn is not None and do_front and not self._batched) else None
        if do_front and not self._batched and self._src_cn is None:     # batched: the shared buffer is cleared once for all clips
            self.tgt_id.zero_()

        for l in range(self.L):
            self._layer = l
            W = self.lw[l]
            Ucat = self.Ucat0 if l == 0 else self.Ucat
            curK = self.bankK[l][cur]
            curV = self.bankV[l][cur]
            seg_a = do_front if l == 0 else do_rest      # norms, projections of this layer
            seg_b = do_rest                              # from the memory reads on
            self._forward_layer(l, W, Ucat, curK, curV, ref_frame, seg_a, seg_b, T, rows, map_bank, map_short)
        if do_rest:
            # -- final GroupNorm over [tgt | tgt_id] (transformer.py:806-808)
            # (one-clip engines: the last layer's projection left split-K partials -- folded here by the statistics pass
            # instead of by a LayerNorm launch whose planes nobody read)
            fold = self._gn_fold
            with self._ev("gn2_stats_kernel + gn2_apply_kernel", nbytes=N * 512 * 4.0 * (3 + (self.KS if fold else 0))):
                hip.check(lib.rmem_groupnorm2_fold(self._tg.data_ptr(), self._tgi.data_ptr(),
                                                   self.parts.data_ptr() if fold else None, self.KS if fold else 0, N * 512, 512,
                                                   N, 256, self.gn_gamma.data_ptr(), self.gn_beta.data_ptr(), 1e-5,
                                                   self.gn_ws.data_ptr(), self.out.data_ptr(), 512, st),
                          "rmem_groupnorm2_fold")

    def _forward_layer(self, l, W, Ucat, curK, curV, ref_frame, seg_a, seg_b, T, rows, map_bank, map_short):
        N, Np, ns = self.N, self.Npad, self.nsplit
        lib = hip.load()
        cur = self.cur
        if seg_a:
            # -- norms + projections (transformer.py:1104-1123); the norms first fold in the
            #    split-K partials of the previous layer's self-attention projection
            if l > 0:
                self._ln2(W.ln1, self.x_pl, 256, 0, W.lnid1, self.z_pl[l], 256, 0, parts=True)
            elif getattr(self, "_src_cn", None) is not None:
                src = self._src_cn
                hip.check(lib.rmem_layernorm_cn(src.data_ptr(), src.stride(0), self.tgt.data_ptr(), self.tgt_id.data_ptr(),
                                                W.ln1[0].data_ptr(), W.ln1[1].data_ptr(), N, 256, 1e-5,
                                                self.x_pl.hi.data_ptr(), self.x_pl.lo.data_ptr(), 256, hip.stream_ptr()),
                          "rmem_layernorm_cn")
            else:
                self._ln(self.tgt, W.ln1, self.x_pl, 256, parts_col=None)
            pe = W.pe_x[T]
            grp = [
                hip.linear(self.x_pl, W.Wq, N, 128, 256, ldx=256, ldy=256, bias=W.bq, pa=curK, ldpa=128,
                           pb=self.Qpe, ldpb=128, addvec=self.cur_pe, nsplit=ns, tile=self._tile(64), launch=False),
                # relative-position bias of the windowed read, anti-diagonal layout (rmem_read_args.rcs)
                hip.linear(self.x_pl, W.Wrel_x, N, self.WIN, 256, ldx=256, ldy=256, bias=W.brel_x,
                           d0=self.R.data_ptr(), ldd0=self.ldr, d0_cs=self.rcs, nsplit=ns, tile=self._tile(64), launch=False),
                # temporal-PE bias of the long-term read: bias_pe[q][t] = (Q[q] + cur_pe) . mem_pe[row(t, T)]
                hip.linear(self.x_pl, pe[0], N, T, 256, ldx=256, ldy=256, bias=pe[1],
                           d0=self.bias_pe.data_ptr(), ldd0=T, nsplit=ns, tile=self._tile(64), launch=False),
                # V = silu(linear_V(x)) -> columns 0..511 of the current slot's blocked-16 V planes
                hip.linear(self.x_pl, W.Wv, N, 512, 256, ldx=256, ldy=256, bias=W.bv, act=1,
                           pa=curV, ldpa=1024, pa_blocked=True, nsplit=ns, tile=self._tile(64), launch=False),
                hip.linear(self.x_pl, W.Wu, N, 512, 256, ldx=256, ldy=256, bias=W.bu, act=1,
                           d0=Ucat.data_ptr(), ldd0=1024, nsplit=ns, tile=self._tile(64), launch=False)]
            if l > 0:
                grp.append(hip.linear(self.z_pl[l], W.Widu, N, 512, 256, ldx=256, ldy=256, bias=W.bidu, act=1,
                                      d0=Ucat.data_ptr() + 512 * 4, ldd0=1024, nsplit=ns, tile=self._tile(64),
                                      launch=False))
            cols = 128 + self.WIN + T + 512 + 512 + (512 if l > 0 else 0)
            fl_front = 2.0 * N * 256 * cols
            by_front = 4.0 * (N * 256 * (2 if l > 0 else 1) + cols * 256 + N * cols)
            with self._ev("linear_stream_kernel (projections)", flops=fl_front, nbytes=by_front):
                hip.linear_grouped(grp)
            if ref_frame:
                self._idv(l, cur)
        early = (l == 0 and self._split_parts and self.early_long_read and not ref_frame and self.branch_order == "serial"
                 and not self._batched)
        if early and seg_a:
            A = self._read_args(self.ws_main, 0, T, self.bankK[l], self.bankV[l], map_bank, self.Qpe,
                                self.bias_pe, Ucat, True, self.ks_long, uneven=True)
            hip.check(lib.rmem_attn_read(C.byref(A[0]), hip.stream_ptr()), "rmem_attn_read")
        if not seg_b:
            return
        # -- long-term memory read (transformer.py:1140-1192, attention.py:174-209) and short-term
        #    windowed read (transformer.py:1199, attention.py:289-358): independent until the
        #    projection; ONE fused-read launch, one combine and one depth-wise-conv launch for both
        #    ("serial_unpaired": one launch each, bit-identical, kept for the equivalence test)
        A = self._read_args(self.ws_main, 0, T, self.bankK[l], self.bankV[l], map_bank, self.Qpe,
                            self.bias_pe, Ucat, l == 0, self.ks_long, uneven=True)
        B = self._read_args(self.ws_side, 1, 1, self.bankK[l], self.bankV[l], map_short, curK, None,
                            Ucat, False, self.ks_win)
        if self.branch_order == "serial":
            self._read_pair(A, B, long_done=early)
            with self._ev("dwconv5x5_split_kernel", nbytes=2 * N * 1024 * 8.0):
                hip.check(lib.rmem_dwconv5x5_split2(
                    self.ws_main.G.data_ptr(), self.ws_side.G.data_ptr(), 1024, W.dw_lt.data_ptr(),
                    W.dw_st.data_ptr(), self.h, self.w, 1024, self.Ylt.hi.data_ptr(), self.Ylt.lo.data_ptr(),
                    self.Yst.hi.data_ptr(), self.Yst.lo.data_ptr(), 1024, hip.stream_ptr()), "rmem_dwconv5x5_split2")
        else:
            self._read(B)
            self._dwconv(self.ws_side, W.dw_st, self.Yst)
            self._read(A)
            self._dwconv(self.ws_main, W.dw_lt, self.Ylt)
        # -- both projections (transformer.py:1212-1220) as ONE split-K GEMM; the residual
        #    adds happen in the norms that follow (rmem_layernorm_red)
        with self._ev("linear_stream_kernel (projections)", flops=2.0 * N * 2048 * 512,
                      nbytes=4.0 * (N * 2048 + 512 * 2048 + N * 512), overhead=4.0 * (self.KS - 1) * N * 512):
            hip.linear(self.Ylt, W.Wp_ls, N, 512, 2048, ldx=1024, ldy=2048, x2=self.Yst, ldx2=1024,
                       kx_split=1024, bias=W.bp_ls, nsplit=ns, tile=self._tile(192), ksplits=self.KS, parts=self.parts,
                       part_stride=N * 512)       # 64 x 128 tiles: 27.5 -> 21.2 us (L2 -> LDS traffic -25 %)
        # -- gated self attention (transformer.py:1223-1232, attention.py:151-209)
        self._ln2(W.ln2, self.s_pl, 512, 0, W.lnid2, self.s_pl, 512, 256, parts=True)
        sQK = Planes(self.selfQK.hi[0], self.selfQK.lo[0])
        grp = [
            hip.linear(self.s_pl, W.Wqk, N, 128, 512, ldx=512, ldy=512, bias=W.bqk, pa=sQK, ldpa=128,
                       nsplit=ns, tile=self._tile(64), launch=False),
            # V = silu([V1(s[:256]) | V2(s[256:])]) -> blocked-16 planes (two diagonal blocks)
            hip.linear(self.s_pl, W.Wv12, N, 512, 256, ldx=512, ldy=256, bias=W.bv12, act=1,
                       pa=self.selfV, ldpa=1024, pa_blocked=True, nbatch=2, bsx=256, bsy=512 * 256, bsbias=512,
                       bspa=512 * 16, nsplit=ns, tile=self._tile(64), launch=False),
            hip.linear(self.s_pl, W.Wu12, N, 512, 256, ldx=512, ldy=256, bias=W.bu12, act=1,
                       d0=self.Uself.data_ptr(), ldd0=1024, nbatch=2, bsx=256, bsy=512 * 256,
                       bsbias=512, bsd=512, nsplit=ns, tile=self._tile(64), launch=False)]
        with self._ev("linear_stream_kernel (projections)", flops=2.0 * N * (512 * 128 + 4 * 256 * 512),
                      nbytes=4.0 * (N * 512 + 128 * 512 + 4 * 512 * 256 + N * (128 + 2048))):
            hip.linear_grouped(grp)
        self._read(self._read_args(self.ws_main, 0, 1, self.selfQK, self.selfV, None, sQK, None, self.Uself,
                                   False, self.ks_self))
        self._dwconv(self.ws_main, W.dw_self, self.Ylt)
        if l + 1 < self.L:     # split-K, folded into the next layer's norm1 / id_norm1
            with self._ev("linear_stream_kernel (projections)", flops=2.0 * N * 1024 * 512,
                          nbytes=4.0 * (N * 1024 + 512 * 1024 + N * 512), overhead=4.0 * (self.KS - 1) * N * 512):
                hip.linear(self.Ylt, W.Wp_self, N, 512, 1024, ldx=1024, ldy=1024, bias=W.bp_self, nsplit=ns,
                           tile=self._tile(192), ksplits=self.KS, parts=self.parts, part_stride=N * 512)
        elif self.clips_per_launch == 1 and not self._force_tiles:
            # last layer: the same split-K launch; its partials are folded into tgt / tgt_id by the LayerNorm kernel
            # (planes into the free self-attention input buffer, unused) -- 12.9 + 6.3 us against 24.7 us for the
            # read-modify-write epilogue of a full-K launch -- and the GroupNorm reads tgt / tgt_id
            with self._ev("linear_stream_kernel (projections)", flops=2.0 * N * 1024 * 512,
                          nbytes=4.0 * (N * 1024 + 512 * 1024 + N * 512), overhead=4.0 * (self.KS - 1) * N * 512):
                hip.linear(self.Ylt, W.Wp_self, N, 512, 1024, ldx=1024, ldy=1024, bias=W.bp_self, nsplit=ns,
                           tile=self._tile(192), ksplits=self.KS, parts=self.parts, part_stride=N * 512)
            self._gn_fold = True           # the GroupNorm's statistics pass folds these partials (rmem_groupnorm2_fold)
        else:                  # (recorded launches) last layer: accumulate straight into tgt / tgt_id
            hip.linear(self.Ylt, W.Wp_self, N, 512, 1024, ldx=1024, ldy=1024, bias=W.bp_self,
                       d0=self.tgt.data_ptr(), ldd0=256, d1=self.tgt_id.data_ptr(), ldd1=256,
                       csplit=256, accumulate=True, nsplit=ns)

    # ------------------------------------------------------------------ memory update
    def update_short_memories(self, update_long: bool, frame_index: int = 0):
        """update_short_memories + update_long_term_memory (transformer.py:826-878).
        ``assign_identity`` must have been called with the current mask.  `frame_index` = the frame step the
        engine appends to long_memories_indexes (aot_engine.py:341-343): rmem_bank_state.index keeps the same id."""
        self._update_device(update_long)
        self._update_host(update_long, frame_index)

    def update_key(self, update_long: bool):
        return (self.cur,)

    def _update_device(self, update_long: bool):
        """Capturable device part: the three ID_V GEMMs into the current slot (one launch)."""
        hip.linear_grouped([self._idv(l, self.cur, launch=False) for l in range(self.L)])

    def _update_host(self, update_long: bool, frame_index: int = 0):
        self.short = self.cur
        if update_long:
            self.resolve_policy(block=True)            # (an eviction from `gap` frames ago: long since there)
            self.bank = self.bank + [self.cur]
            if self._dev_policy:
                hip.check(hip.load().rmem_bank_append(self.maps.data_ptr(), self.bank_state.data_ptr(), self.cur,
                                                      int(frame_index), hip.stream_ptr()), "rmem_bank_append")

    def restrict_long_memories(self, indexes: List[int], fg: Optional[torch.Tensor] = None,
                               logits: Optional[torch.Tensor] = None) -> Optional[int]:
        """restrict_long_memories (transformer.py:880-991).  Give either fg ([N] fp32 on device) or the decoder
        logits [1,C,Hl,Wl] the foreground weights are taken from (aot_engine.py:350-356).  Mutates ``indexes``
        like the reference.  Host rule: returns the dropped position or None.  Device rule (the default for one
        clip per launch): everything is queued on the stream, nothing is read back here -- returns None, and
        ``indexes`` / self.bank catch up in resolve_policy()."""
        if logits is not None:
            lg = logits[0].contiguous()
            hip.check(hip.load().rmem_fg_weights(lg.data_ptr(), lg.shape[0], lg.shape[1], lg.shape[2], self.h, self.w,
                                                 self.fg.data_ptr(), hip.stream_ptr()), "rmem_fg_weights")
            fg = self.fg
        self._mass_reduce_device(fg)
        if not self._dev_policy:
            w = self.w_out[:self.mass_T].cpu().numpy().astype(np.float32)        # the one D2H per long update
            return self._restrict_host(indexes, w)
        return self._policy_device(indexes)

    def _policy_device(self, indexes: List[int]) -> None:
        """The EMA + UCB rule and the deletion of the dropped slot, queued on the stream after the mass reduction;
        `indexes` and self.bank catch up in resolve_policy()."""
        self._policy_seq += 1
        hip.check(hip.load().rmem_bank_policy_step(
            self.maps.data_ptr(), self.bank_state.data_ptr(), self.w_out.data_ptr(), self.mass_T, self.cap,
            self.cfg.FORMER_MEM_LEN, self.policy_result.data_ptr(), hip.stream_ptr()), "rmem_bank_policy_step")
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._pending = dict(event=ev, seq=self._policy_seq, indexes=indexes, expect_drop=len(self.bank) > self.cap)
        return None

    def _mass_reduce_device(self, fg: torch.Tensor):
        hip.check(hip.load().rmem_attn_mass_reduce(self.mass.data_ptr(), self.N, self.mass_T, fg.data_ptr(),
                                                   self.w_out.data_ptr(), hip.stream_ptr()),
                  "rmem_attn_mass_reduce")

    def _restrict_host(self, indexes: List[int], w: np.ndarray) -> Optional[int]:
        w = w / w.sum(dtype=np.float32)
        drop, self.ema, self.visits = rmem_policy_step(w, indexes, self.ema, self.visits,
                                                       self.cfg.FORMER_MEM_LEN)
        self.last_policy = dict(w=w.tolist(), drop=drop)
        if len(self.bank) > self.cap:
            del self.bank[drop]
            indexes.remove(indexes[drop])
            return drop
        return None
