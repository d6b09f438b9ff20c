"""Inference engines with the reference's API, backed by the HIP LSTT.

Drop-in for /root/reference/aot_plus/networks/engines:
``build_engine(name, phase='eval', aot_model=..., gpu_id=..., long_term_mem_gap=...)``
returns an object exposing ``restart_engine / add_reference_frame /
match_propogate_one_frame / update_memory``, the writable ``long_term_mem_gap`` and the
read attributes ``input_size_2d / enc_size_2d / enc_hw`` -- the surface
``managers/evaluator.py:344-523`` and ``tools/demo.py:113-190`` drive.

Encoder and FPN decoder run through PyTorch-ROCm/MIOpen unchanged; everything between
them (LSTT, ID assignment, memory bank, RMem eviction) is rmem_amd.lstt / csrc.
"""
from __future__ import annotations

import os
import queue
import threading
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .lstt import DeAOTLSTT
from .lstt_aot import AOTLSTT


def default_lookahead() -> int:
    """Frames worth announcing through `next_img`: 2b-1 for an encoder batch b > 1 (RMEM_ENC_BATCH,
    default 2), else 2 -- see DeAOTEngine._prefetch."""
    b = max(1, int(os.environ.get("RMEM_ENC_BATCH", "2")))
    return 2 * b - 1 if b > 1 else 2


class _GraphLauncher(threading.Thread):
    """Host helper thread that launches a hipGraph on a given stream (see DeAOTEngine._prefetch)."""

    def __init__(self, device):
        super().__init__(daemon=True, name="rmem-graph-launcher")
        self.device = device
        self.jobs: "queue.Queue" = queue.Queue()
        self.error = None
        self.start()

    def submit(self, stream, after, dst, srcs, graph, done_event) -> threading.Event:
        """dst: the graph's static input [b,3,H,W]; srcs: b images [1,3,H,W] (a short list is padded
        with its last image: the surplus outputs are never read)."""
        launched = threading.Event()
        self.jobs.put((stream, after, dst, list(srcs), graph, done_event, launched))
        return launched

    def run(self):
        torch.cuda.set_device(self.device)
        while True:
            stream, after, dst, srcs, graph, done_event, launched = self.jobs.get()
            try:
                with torch.no_grad(), torch.cuda.stream(stream):
                    stream.wait_event(after)
                    for i in range(dst.shape[0]):
                        src = srcs[min(i, len(srcs) - 1)]
                        dst[i:i + 1].copy_(src, non_blocking=True)
                        src.record_stream(stream)     # the caller may free it before this copy has run
                    graph.replay()
                    done_event.record(stream)
            except BaseException as e:          # surfaced by the next wait on the engine thread
                self.error = e
            finally:
                launched.set()


class DeAOTEngine(nn.Module):
    """One engine = up to MODEL_MAX_OBJ_NUM objects (engines/aot_engine.py:18-568); serves both
    the DeAOT (rmem_amd.lstt.DeAOTLSTT) and the AOT (rmem_amd.lstt_aot.AOTLSTT) models."""

    def __init__(self, aot_model, gpu_id=0, long_term_mem_gap=9999, short_term_mem_skip=1,
                 nsplit: int = 3, use_graphs: Optional[bool] = None):
        super().__init__()
        if use_graphs is None:
            use_graphs = os.environ.get("RMEM_NO_GRAPHS") is None
        self.use_graphs = bool(use_graphs)
        self._fg, self._ug = {}, {}          # hipGraphs: frame graphs / update graphs by key
        self._tg = {}                        # `tail` graphs by lstt.graph_key() (sampled kernel timing, see _graphed_frame)
        self._dg = {}                        # decoder (+ upsample) graphs by (output size, shape, feature copy, obj_nums)
        self._eg = {}                        # encoder graphs by (img shape, parity): (graph, static img, features)
        self._g_lab = {}                     # static graph inputs per shape (graphs keep their address)
        # Encoder feature copies.  A copy `par` belongs to a group = one encoder hipGraph: three
        # single-frame groups (pars 0-2) and two groups of `_enc_batch` frames each (one pass over a
        # [b,3,H,W] batch: the encoder's kernels are under-filled at batch 1 -- 1.27 ms per frame
        # against 1.06 at b=2 and 0.95 at b=4 -- and the label maps do not move: 31 mismatching
        # pixels over the golden 480p clip at b=1 and b=2, 33 at b=4).
        self._par = 0                        # copy that holds the current frame
        self._pending = []                   # prefetched, not yet consumed: [(image identity, copy, launched)]
        self._enc_batch = max(1, int(os.environ.get("RMEM_ENC_BATCH", "2")))
        b = self._enc_batch
        self._groups = [[0], [1], [2]] + ([[3 + g * b + i for i in range(b)] for g in range(2)] if b > 1 else [])
        self._group_of = {par: g for g, pars in enumerate(self._groups) for par in pars}
        self._feats = {}                     # (shape, par) -> FeatureList view of that copy
        self._enc_warm = set()               # (shape, batch) whose convolutions MIOpen has already seen
        self._hoist = None                   # hoisted front part of the next frame's LSTT in flight (see _try_hoist)
        self._hoist_stream = None
        self._gen = 0                        # bumped by whatever invalidates a hoisted front (reference frame, restart)
        self.hoist_enabled = os.environ.get("RMEM_HOIST", "1") == "1"
        self._enc_stream = None
        self._enc_done = None
        self._launcher = None
        self._eager_frames = 0
        self._geoms = []                     # (output size, image shape) with live graphs, least recently used first
        self.prefetch_at = os.environ.get("RMEM_PREFETCH_AT", "lstt")   # decoder | lstt
        if short_term_mem_skip != 1:
            raise NotImplementedError("short_term_mem_skip != 1 (reference evaluator always uses 1)")
        self.cfg = aot_model.cfg
        self.align_corners = self.cfg.MODEL_ALIGN_CORNERS
        self.AOT = aot_model
        self.max_obj_num = aot_model.max_obj_num
        self.gpu_id = gpu_id
        self.long_term_mem_gap = long_term_mem_gap
        self.nsplit = nsplit
        self.lstt: Optional[DeAOTLSTT] = None
        self._lstt_wv = 0
        self.restart_engine()

    @property
    def long_memories_indexes(self) -> List[int]:
        """Frame indexes of the bank slots (aot_engine.py:322, 346).  With the eviction rule on the device the
        host's copy may lag by one decision; reading it waits for that decision."""
        l = self.__dict__.get("_modules", {}).get("lstt") or self.__dict__.get("lstt")
        if l is not None and hasattr(l, "resolve_policy"):
            l.resolve_policy(block=True)
        return self._lmi

    @long_memories_indexes.setter
    def long_memories_indexes(self, v):
        self._lmi = v

    def restart_engine(self):                                   # aot_engine.py:533-563
        self.frame_step = 0
        self.last_mem_step = -1
        self.obj_nums = None
        self.enc_size_2d = None
        self.enc_hw = None
        self.input_size_2d = None
        self.long_memories_indexes: List[int] = []
        self.pred_id_logits = None
        self._drop_pending()
        self._drop_hoist()
        # Every clip follows the same schedule from its first frame (two eagerly issued frames, then
        # hipGraph replay with encoder prefetch): which frames share a batched encoder pass decides
        # their MIOpen rounding (batch 1 and batch 2 pick different algorithms, 1e-6 on the features),
        # so a schedule that depended on what the engine ran before would make a clip's label maps depend
        # on the clips before it.  (MIOpen itself is not bit-reproducible between processes:
        # tools/clip_determinism_probe.py gives one of two label sequences for a clip with a near-tie
        # pixel, from identical commands.)
        self._eager_frames = 0
        self._par = 0
        if self.lstt is not None and self._lstt_wv != self.AOT.__dict__.get("_weights_version", 0):
            self.lstt = None                 # weights were (re)loaded: re-pack at the next reference frame
            self._fg, self._ug, self._dg, self._tg = {}, {}, {}, {}
            self._geoms = []
        if self.lstt is not None:
            # graph-pointer audit: the captured graphs hold raw addresses of the LSTT's buffers
            sig = self.lstt.buffer_signature() if hasattr(self.lstt, "buffer_signature") else None
            if getattr(self, "_graph_sig", sig) != sig:
                import warnings
                warnings.warn("rmem_amd: an LSTT buffer was reallocated since its hipGraphs were captured; dropping the graphs")
                self._fg, self._ug, self._dg, self._tg = {}, {}, {}, {}
                self._drop_pending()
                self._drop_hoist()
            self._graph_sig = sig
            self.lstt.clear_memory()

    def update_size(self, input_size, enc_size):                # aot_engine.py:565-568
        self.input_size_2d = tuple(int(v) for v in input_size)
        self.enc_size_2d = tuple(int(v) for v in enc_size)
        self.enc_hw = self.enc_size_2d[0] * self.enc_size_2d[1]
        wv = self.AOT.__dict__.get("_weights_version", 0)
        if self.lstt is None or (self.lstt.h, self.lstt.w) != self.enc_size_2d or self._lstt_wv != wv:
            self._lstt_wv = wv          # load_network() after this engine was built: packed weights are stale
            dev = next(self.AOT.parameters()).device
            cls = DeAOTLSTT if self.cfg.MODEL_VOS == "deaot" else AOTLSTT
            self.lstt = cls(self.AOT, self.enc_size_2d[0], self.enc_size_2d[1], dev, self.nsplit)
            self._fg, self._ug, self._dg, self._tg = {}, {}, {}, {}      # graphs hold pointers into the old LSTT buffers
            self._geoms = []
            self._drop_pending()
            self._drop_hoist()
            self._eg, self._g_lab, self._feats = {}, {}, {}

    def _label_u8(self, mask: torch.Tensor) -> torch.Tensor:
        """[1,1,H,W] (or [1,H,W]) label ids -> contiguous uint8 [H,W] on device."""
        m = mask
        while m.dim() > 2:
            m = m[0]
        return m.to(torch.uint8).contiguous()

    def _tokens(self, enc_last: torch.Tensor) -> torch.Tensor:
        """bchw_2_lbc (utils/tensor.py:3-6) for batch 1: [1,C,h,w] -> [N,C]."""
        return enc_last[0].flatten(1).t().contiguous()

    @torch.no_grad()
    def add_reference_frame(self, img=None, mask=None, frame_step=-1, obj_nums=None, img_embs=None):
        """aot_engine.py:241-325."""
        if self.obj_nums is None and obj_nums is None:
            raise ValueError("No objects for reference frame!")
        if obj_nums is not None:
            self.obj_nums = obj_nums
        if frame_step == -1:
            frame_step = self.frame_step
        if mask is None:
            raise ValueError("No mask for reference frame!")
        self._drop_hoist()
        enc = self.AOT.encode_image(img) if img_embs is None else img_embs
        if enc is None:
            raise ValueError("No image for reference frame!")
        if self.input_size_2d is None:
            self.update_size(img.shape[2:], enc[-1].shape[2:])
        # no ignore channel on reference frames: the reference calls assign_identity without an
        # ignore mask here (aot_engine.py:304 -> :209-213), so a 255 pixel contributes nothing
        self.lstt.assign_identity(self._label_u8(mask), ignore=False)
        self.lstt.ref_frame_index = self.frame_step
        out = self.lstt.forward(self._tokens(enc[-1]), ref_frame=True)
        self.last_mem_step = frame_step
        # A reference frame re-initialises the bank to one slot.  The reference keeps the old
        # frame indexes (aot_engine.py:322-323) and then raises at the next long-term update
        # (transformer.py:954, size mismatch); here the index list restarts with the bank.
        self.long_memories_indexes = [self.frame_step]
        self.decode_current_logits(enc, out)

    @torch.no_grad()
    def match_propogate_one_frame(self, img=None, img_embs=None, mask=None, output_size=None, next_img=None):
        """aot_engine.py:398-436.  `next_img` (extension, optional): the frame -- or a list of the
        next frames, in order -- that will be passed to the NEXT call(s); their encoder passes
        then run on a second stream concurrently with this frame's LSTT / decoder (the encoder
        does not depend on the memory)."""
        self.frame_step += 1
        if self._graph_ok(img, img_embs):
            return self._graphed_frame(img, output_size, next_img)
        self._eager_frames += 1
        self._drop_hoist()
        enc = img_embs
        if enc is None:
            enc = self._take_prefetched(img)
        if enc is None:
            enc = self.AOT.encode_image(img)
        out = self.lstt.forward(self._tokens(enc[-1]), ref_frame=False)
        return self.decode_current_logits(enc, out, output_size)

    # ------------------------------------------------------------------ hipGraph replay
    # One frame = encoder -> LSTT -> decoder -> upsample is ~270 launches; issued eagerly the host
    # needs as long as the GPU.  In steady state a frame is replayed from
    # two hipGraphs: the encoder graph (depends on the image shape only; two copies with their
    # own feature buffers, so that the next frame's encoder can run on a second stream while
    # this frame's LSTT and decoder read the other copy) and the frame graph (LSTT + decoder +
    # upsample).  A frame graph bakes in everything that is not read from device memory: the
    # slot the frame is written to, the bank depth T, the tensor shapes and the feature copy --
    # that tuple is the cache key; the logical->physical slot map stays in a device array, so
    # appends / evictions never invalidate a graph.
    def _graph_ok(self, img, img_embs) -> bool:
        return (self.use_graphs and img_embs is None and img is not None and img.is_cuda
                and self._eager_frames >= 2 and not self.lstt._timing)

    class _ImgId:
        """Identity of an announced frame: the tensor OBJECT (a strong reference, so its address
        cannot be recycled for another frame while the entry is pending) and its version counter.
        Callers must pass to match_propogate_one_frame the same tensor objects they announced."""
        __slots__ = ("img", "version")

        def __init__(self, img):
            self.img, self.version = img, img._version

        def __eq__(self, other):
            return isinstance(other, DeAOTEngine._ImgId) and self.img is other.img and self.version == other.version

        def __hash__(self):
            return hash((id(self.img), self.version))

    @classmethod
    def _img_id(cls, img):
        return cls._ImgId(img)

    @property
    def lookahead(self) -> int:
        """How many upcoming frames a caller should announce (`next_img`) for the prefetch to run
        whole batches without ever stalling a frame: 2b-1 (b = encoder batch), 2 at b = 1."""
        return 2 * self._enc_batch - 1 if self._enc_batch > 1 else 2

    def _encoder_graph(self, shape, group, like):
        """hipGraph of the encoder pass of feature group `group` (static input [b,3,H,W])."""
        ent = self._eg.get((shape, group))
        if ent is None:
            from .model import FeatureList
            torch.cuda.synchronize()
            pars = self._groups[group]
            g_img = torch.zeros((len(pars),) + tuple(like.shape[1:]), dtype=like.dtype, device=like.device)
            if (shape, len(pars)) not in self._enc_warm:
                # MIOpen picks (and may compile) its solvers at the first call of every convolution
                # shape, which is not possible inside a capture: one eager pass per batch size first
                self.AOT.encode_image(g_img)
                torch.cuda.synchronize()
                self._enc_warm.add((shape, len(pars)))
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                enc = self.AOT.encode_image(g_img)
            ent = self._eg[(shape, group)] = (g, g_img, enc)
            for i, par in enumerate(pars):       # per-frame views (batch is the outermost axis: contiguous)
                one = FeatureList([x[i:i + 1] for x in enc])
                if getattr(enc, "adapters", None) is not None:
                    one.adapters = [x[i:i + 1] for x in enc.adapters]
                self._feats[(shape, par)] = one
            if self._enc_stream is None:
                from .streams import concurrent_stream
                self._enc_stream = concurrent_stream(like.device)
                self._enc_done = [torch.cuda.Event() for _ in self._groups]
        return ent

    def _features(self, shape, par, like):
        self._encoder_graph(shape, self._group_of[par], like)
        return self._feats[(shape, par)]

    def _drop_pending(self):
        """Forget prefetched frames (restart / resize).  Whatever touches their feature copies next
        must still come after the encoder stream is done with them."""
        pend, self._pending = getattr(self, "_pending", []), []
        for _, par, launched in pend:
            launched.wait()
            torch.cuda.current_stream().wait_event(self._enc_done[self._group_of[par]])

    def _take_prefetched(self, img):
        """Features of `img` if an earlier call prefetched its encoder pass, else None.  Frames
        are consumed in the order they were announced; older pending entries are discarded."""
        if not self._pending:
            return None
        ident = self._img_id(img) if (img is not None and img.is_cuda) else None
        hit = next((i for i, p in enumerate(self._pending) if p[0] == ident), None)
        if hit is None:
            self._drop_pending()
            return None
        drop, (_, par, launched), self._pending = self._pending[:hit], self._pending[hit], self._pending[hit + 1:]
        for _, p, ln in drop + [(None, par, launched)]:
            ln.wait()                            # the helper thread has queued the pass and its event
            if self._launcher.error is not None:
                raise RuntimeError("encoder prefetch failed") from self._launcher.error
            torch.cuda.current_stream().wait_event(self._enc_done[self._group_of[p]])
        self._par = par
        return self._feats[(tuple(img.shape), par)]

    def _prefetch(self, next_imgs, shape):
        """Encoder passes of the announced next frames into free feature groups on the encoder
        stream.  Each pass is ordered after everything already queued on the current stream (the
        previous reader of those copies) and after the passes queued before it, concurrent with
        whatever is queued next.  Whole batches of `_enc_batch` not-yet-encoded frames go through
        one pass; a single-frame pass is used only when the very next frame would otherwise not be
        ready.  With `lookahead` frames announced a batch pass is queued two calls before its
        first frame is needed, so the encoder stream always has work and also fills the decoder /
        label / memory-update phase of a frame, whose small kernels leave most of the GPU idle.
        The graphs are launched from a helper host thread: hipGraphLaunch enqueues node by node
        (~8 us of host time per kernel), so launching an encoder graph and then the frame graphs
        from one thread leaves the GPU waiting for the host; replay() releases the GIL, the
        launches proceed in parallel."""
        if next_imgs is None:
            return
        if isinstance(next_imgs, torch.Tensor):
            next_imgs = [next_imgs]
        ann = []
        for img in list(next_imgs)[:max(self.lookahead, 2)]:
            if img is None or not img.is_cuda or tuple(img.shape) != shape:
                break
            ann.append(img)
        b = self._enc_batch
        while True:
            have = {p[0] for p in self._pending}
            todo = [im for im in ann if self._img_id(im) not in have]
            if not todo:
                return
            busy = {self._par} | {p[1] for p in self._pending}
            free = [g for g, pars in enumerate(self._groups) if not (set(pars) & busy)]
            batch_groups = [g for g in free if len(self._groups[g]) > 1]
            single_groups = [g for g in free if len(self._groups[g]) == 1]
            if b > 1 and len(todo) >= b and batch_groups:
                group, imgs = batch_groups[0], todo[:b]
            elif todo[0] is ann[0] and single_groups:          # the next frame itself is not encoded yet
                group, imgs = single_groups[0], todo[:1]
            elif todo[0] is ann[0] and batch_groups:
                group, imgs = batch_groups[0], todo[:b]         # short batch, padded by the launcher
            else:
                return
            ent = self._encoder_graph(shape, group, imgs[0])
            after = torch.cuda.Event()
            after.record(torch.cuda.current_stream())
            if self._launcher is None:
                self._launcher = _GraphLauncher(imgs[0].device)
            launched = self._launcher.submit(self._enc_stream, after, ent[1], imgs, ent[0], self._enc_done[group])
            for im, par in zip(imgs, self._groups[group]):
                self._pending.append((self._img_id(im), par, launched))

    # -- hoisted front part of the next frame's LSTT.  Layer 0 up to and including the score passes
    #    needs the next frame's encoder features (prefetched), the bank keys and the slot maps, not
    #    this frame's label: on frames whose memory update will not touch the long-term bank it is
    #    issued on a third stream right after this frame's LSTT, beside the decoder, the label
    #    kernels and the memory update, whose small kernels leave the GPU mostly idle.
    def _drop_hoist(self):
        h, self._hoist = getattr(self, "_hoist", None), None
        self._gen = getattr(self, "_gen", 0) + 1
        if h is not None:
            torch.cuda.current_stream().wait_event(h["event"])

    def _try_hoist(self, next_img, shape, osz):
        l = self.lstt
        if not self.hoist_enabled or next_img is None or getattr(l, "branch_order", "") != "serial":
            return
        nxt = next_img if isinstance(next_img, torch.Tensor) else (next_img[0] if len(next_img) else None)
        if nxt is None or not nxt.is_cuda or tuple(nxt.shape) != shape:
            return
        if (not self.cfg.NO_LONG_MEMORY) and self.frame_step - self.last_mem_step >= self.long_term_mem_gap:
            return                                        # this frame's update appends to / evicts from the bank
        ident = self._img_id(nxt)
        pend = next((p for p in self._pending if p[0] == ident), None)
        if pend is None:
            return
        par, cur = pend[1], l.next_free_slot()
        ent = self._fg.get(((l._T, cur), osz, shape, par, tuple(self.obj_nums)))
        if ent is None:
            return                                        # captured when a frame first runs with that key
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())            # this frame's LSTT is done
        if self._hoist_stream is None:
            from .streams import concurrent_stream
            self._hoist_stream = concurrent_stream(nxt.device)
        pend[2].wait()                                    # the encoder pass and its event are queued
        done = torch.cuda.Event()
        with torch.cuda.stream(self._hoist_stream):
            self._hoist_stream.wait_event(ev)
            self._hoist_stream.wait_event(self._enc_done[self._group_of[par]])
            ent[4].replay()
            done.record(self._hoist_stream)
        self._hoist = dict(ident=ident, cur=cur, T=l._T, par=par, bank=tuple(l.bank), short=l.cur,
                           gen=self._gen, event=done)

    def _graphed_frame(self, img, output_size, next_img=None):
        l = self.lstt
        h, self._hoist = self._hoist, None
        if h is not None:                                 # before _prepare rewrites the slot maps it reads
            torch.cuda.current_stream().wait_event(h["event"])
        l._prepare(False)
        osz = tuple(int(v) for v in output_size) if output_size is not None else None
        shape = tuple(img.shape)
        if self._take_prefetched(img) is None:            # features not there yet: encode in line
            busy = {p[1] for p in self._pending}
            self._par = next(c for c in (0, 1, 2) if c not in busy)     # a single-frame copy
            g, g_img, _ = self._encoder_graph(shape, self._group_of[self._par], img)
            g_img.copy_(img)
            g.replay()
        par = self._par                                   # the copy that holds this frame's features
        key = (l.graph_key(), osz, shape, par, tuple(self.obj_nums))     # obj_nums is baked into the decoder graph
        self._touch_geometry(osz, shape)
        ent = self._fg.get(key)
        if ent is None:
            # capture every slot variant for this (T, shapes, feature copy) at once: capture
            # executes nothing, so the memory state is untouched, and no capture lands in a
            # later frame
            torch.cuda.synchronize()
            enc = self._features(shape, par, img)
            saved = {k: getattr(l, k) for k in l.graph_variants()[0]}
            for var in l.graph_variants():
                for k, v in var.items():
                    setattr(l, k, v)
                g, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                fused_in = isinstance(l, DeAOTLSTT) and os.environ.get("RMEM_LN_CN", "1") == "1"          # norm1 of layer 0 reads the feature map itself (rmem_layernorm_cn)
                src_cn = enc[-1][0].flatten(1) if fused_in else None
                with torch.cuda.graph(g):
                    if not fused_in:
                        l.tgt.copy_(enc[-1][0].flatten(1).t())
                        l._forward_device(False)
                    else:
                        l._forward_device(False, src_cn=src_cn)
                gf = gr = None
                if self.hoist_enabled and getattr(l, "branch_order", "") == "serial":
                    gf, gr = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gf):
                        if not fused_in:
                            l.tgt.copy_(enc[-1][0].flatten(1).t())
                            l._forward_device(False, "front")
                        else:
                            l._forward_device(False, "front", src_cn=src_cn)
                    with torch.cuda.graph(gr):
                        l._forward_device(False, "rest")
                # the decoder (+ upsample) graph reads the LSTT's static output buffer and this feature
                # copy only: ONE capture per (output size, shape, copy, obj_nums), shared by every
                # (T, slot) variant (a private copy per variant was ~200 activation pools per geometry)
                dkey = (osz, shape, par, tuple(self.obj_nums))
                dent = self._dg.get(dkey)
                if dent is None:
                    g2 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g2):
                        logits = self.AOT.decode_id_logits(l.out, enc)
                        for batch_idx, obj_num in enumerate(self.obj_nums):
                            logits[batch_idx, (obj_num + 1):] = -1e10
                        up = logits if osz is None else F.interpolate(logits, size=osz, mode="bilinear",
                                                                      align_corners=self.align_corners)
                    dent = self._dg[dkey] = (g2, logits, up)
                g2, logits, up = dent
                self._fg[(l.graph_key(), osz, shape, par, tuple(self.obj_nums))] = (g, logits, up, g2, gf, gr)
            for k, v in saved.items():
                setattr(l, k, v)
            ent = self._fg[key]
        # The frame is two graphs, LSTT and decoder, so that the next frame's encoder pass can be
        # released between them: beside the LSTT it competes with latency-bound kernels for
        # workgroup slots (every kernel of both chains ~2x slower), beside the decoder + label
        # post-processing + memory update it fills a GPU that those small kernels leave idle.
        hoisted = (h is not None and ent[5] is not None and h["ident"] == self._img_id(img) and h["cur"] == l.cur
                   and h["T"] == l._T and h["par"] == par and h["bank"] == tuple(l.bank) and h["short"] == l.short
                   and h["gen"] == self._gen)
        if self.prefetch_at == "lstt":
            self._prefetch(next_img, shape)               # runs beside the LSTT graph below
        if getattr(l, "_sample_read", False) and ent[4] is not None:
            # bench.py's roofline sample: the same replayed frame with the fused read of layer 0 issued on its own
            # between two HIP events -- front graph (unless hoisted), the launch, tail graph
            l._sample_read = False
            tail = self._tail_graph(l)
            if not hoisted:
                ent[4].replay()
            l.launch_read2_layer0()
            tail.replay()
        elif getattr(l, "_sample_kernels", False) and ent[4] is not None:
            # bench.py's per-kernel roofline sample: the `rest` part of this frame issued EAGERLY with HIP events around
            # every launch (lstt._ev).  In steady state the host runs frames ahead of the GPU (graph replays), so these
            # launches queue behind the previous frame's work and start back to back like the replayed ones; the front
            # part stays a graph (or ran hoisted).
            l._sample_kernels = False
            if not hoisted:
                ent[4].replay()
            l._kev = l._kev_store
            l._forward_device(False, "rest")
            l._kev = None
            l._kev_frames += 1
        else:
            (ent[5] if hoisted else ent[0]).replay()      # the front part ran beside the previous decoder
        self._hoist_count = getattr(self, "_hoist_count", 0) + int(hoisted)
        if self.prefetch_at != "lstt":
            self._prefetch(next_img, shape)               # released when the LSTT is done
        self._try_hoist(next_img, shape, osz)
        ent[3].replay()
        l._finish(False)
        self.pred_id_logits = ent[1]
        return ent[2]

    def _tail_graph(self, l):
        """`rest` graph without its first launch (lstt._forward_device(part="tail")), all slot variants captured at
        the first request (capture executes nothing)."""
        g = self._tg.get(l.graph_key())
        if g is None:
            torch.cuda.synchronize()
            saved = {k: getattr(l, k) for k in l.graph_variants()[0]}
            for var in l.graph_variants():
                for k, v in var.items():
                    setattr(l, k, v)
                gt = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gt):
                    l._forward_device(False, "tail")
                self._tg[l.graph_key()] = gt
            for k, v in saved.items():
                setattr(l, k, v)
            g = self._tg[l.graph_key()]
        return g

    def _touch_geometry(self, osz, shape):
        """Graph caches are bounded per (output size, image shape): a dataset whose clips differ in
        their original size would otherwise keep every geometry's frame / decoder graphs (and their
        activation pools) alive.  The least recently used geometry beyond RMEM_GRAPH_GEOMS (default 3)
        is dropped; its encoder graphs go with the last geometry that uses the image shape."""
        g = (osz, shape)
        order = self._geoms
        if order and order[-1] == g:
            return
        if g in order:
            order.remove(g)
        order.append(g)
        limit = max(1, int(os.environ.get("RMEM_GRAPH_GEOMS", "3")))
        while len(order) > limit:
            old = order.pop(0)
            self._drop_hoist()
            self._drop_pending()
            torch.cuda.synchronize()
            self._fg = {k: v for k, v in self._fg.items() if (k[1], k[2]) != old}
            self._tg = {}
            self._dg = {k: v for k, v in self._dg.items() if (k[0], k[1]) != old}
            if not any(o[1] == old[1] for o in order):
                self._eg = {k: v for k, v in self._eg.items() if k[0] != old[1]}
                self._feats = {k: v for k, v in self._feats.items() if k[0] != old[1]}

    def decode_current_logits(self, enc, lstt_out, output_size=None):      # aot_engine.py:438-465
        logits = self.AOT.decode_id_logits(lstt_out, enc)
        for batch_idx, obj_num in enumerate(self.obj_nums):
            logits[batch_idx, (obj_num + 1):] = -1e10
        self.pred_id_logits = logits
        if output_size is not None:
            logits = F.interpolate(logits, size=output_size, mode="bilinear",
                                   align_corners=self.align_corners)
        return logits

    def label_buffer(self, shape, device) -> torch.Tensor:
        """The uint8 [H,W] buffer the memory-update graph reads its label map from.  A caller that
        produces the label on the device (rmem_label_resize_nearest) can write it here and pass it
        to update_memory: the copy into the graph's static input is then skipped."""
        shape = tuple(int(v) for v in shape)
        buf = self._g_lab.get(shape)
        if buf is None:
            buf = self._g_lab[shape] = torch.empty(shape, dtype=torch.uint8, device=device)
        return buf

    @torch.no_grad()
    def update_short_term_memory(self, curr_mask, curr_id_emb=None, step=0):
        """aot_engine.py:327-369."""
        if curr_mask.dim() == 4 and curr_mask.shape[1] > 1:
            raise NotImplementedError("probability masks are a training-only input (aot_engine.py:333-334)")
        update_long = False
        if (not self.cfg.NO_LONG_MEMORY) and \
                self.frame_step - self.last_mem_step >= self.long_term_mem_gap:
            update_long = True
            self.last_mem_step = self.frame_step
        lab = self._label_u8(curr_mask)
        l = self.lstt
        if self.use_graphs and lab.is_cuda and self._eager_frames >= 2 and not l._timing:
            g_lab = self._g_lab.get(tuple(lab.shape))
            if g_lab is None:
                g_lab = self._g_lab[tuple(lab.shape)] = torch.empty_like(lab)
            if lab.data_ptr() != g_lab.data_ptr():        # a caller may have written into label_buffer() directly
                g_lab.copy_(lab)
            key = (l.update_key(update_long), tuple(lab.shape))
            g = self._ug.get(key)
            if g is None:
                torch.cuda.synchronize()
                saved = {k: getattr(l, k) for k in l.graph_variants()[0]}
                for var in l.graph_variants():
                    for k, v in var.items():
                        setattr(l, k, v)
                    for ul in (False, True):
                        k2 = (l.update_key(ul), tuple(lab.shape))
                        if k2 in self._ug:
                            continue
                        gg = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(gg):
                            l.assign_identity(g_lab)
                            l._update_device(ul)
                        self._ug[k2] = gg
                for k, v in saved.items():
                    setattr(l, k, v)
                g = self._ug[key]
            g.replay()
            l._update_host(update_long, self.frame_step)
        else:
            l.assign_identity(lab)
            l._update_device(update_long)
            l._update_host(update_long, self.frame_step)
        if update_long:
            idx = self.long_memories_indexes            # (brings the host's view up to date first)
            idx.append(self.frame_step)
            if getattr(self.lstt, "_dev_policy", False):      # the ONE flag the LSTT itself branches on
                # foreground weights, attention-mass reduction, EMA / UCB rule and the deletion of the dropped slot
                # all on the device (rmem_fg_weights, rmem_attn_mass_reduce, rmem_bank_policy_step): no D2H here
                self.lstt.restrict_long_memories(idx, logits=self.pred_id_logits)
            else:
                lg = F.interpolate(self.pred_id_logits, size=self.enc_size_2d, mode="bilinear",
                                   align_corners=True)
                fg = (1 - torch.softmax(lg, dim=1)[:, 0]).reshape(-1).contiguous()
                self.lstt.restrict_long_memories(idx, fg)


class _SubEngineView:
    """What callers read from `DeAOTInferEngine.aot_engines[i]` when the sub-engines of a many-object clip are the
    slots of ONE rmem_amd.batched.BatchedDeAOTEngine (shared launches of the memory path)."""

    def __init__(self, bat, i: int):
        self.bat, self.i = bat, i

    long_memories_indexes = property(lambda self: self.bat.long_memories_indexes[self.i])
    pred_id_logits = property(lambda self: self.bat.pred_id_logits[self.i:self.i + 1])
    frame_step = property(lambda self: self.bat.frame_steps[self.i])
    last_mem_step = property(lambda self: self.bat.last_mem_steps[self.i])
    input_size_2d = property(lambda self: self.bat.input_size_2d)
    enc_size_2d = property(lambda self: self.bat.enc_size_2d)
    enc_hw = property(lambda self: self.bat.enc_hw)
    obj_nums = property(lambda self: [self.bat.obj_nums[self.i]])

    @property
    def lstt(self):
        return self.bat.lstt.clips[self.i]

    def restart_engine(self):
        if self.i == 0:
            self.bat.restart_engine()


class DeAOTInferEngine(nn.Module):
    """Multi-object wrapper (engines/aot_engine.py:571-725, deaot_engine.py:20-56).

    More than `max_aot_obj_num` objects -> one sub-engine per 10 ids (aot_engine.py:675-702).  On the GPU with the DeAOT
    block the sub-engines are the slots of ONE BatchedDeAOTEngine: the image is encoded once, every launch of the memory
    path serves all sub-engines, the decoder runs at batch = number of sub-engines and the frame replays hipGraphs
    (RMEM_MULTI_ENGINE=serial: one DeAOTEngine after the other, eagerly issued, as before round 4
    -- also what the AOT block and substituted engines use)."""

    supports_prefetch = True      # match_propogate_one_frame accepts next_img

    @property
    def lookahead(self) -> int:
        """Number of upcoming frames worth announcing through `next_img` (see DeAOTEngine.lookahead)."""
        return default_lookahead()

    def __init__(self, aot_model, gpu_id=0, long_term_mem_gap=9999, short_term_mem_skip=1,
                 max_aot_obj_num=None, nsplit: int = 3, fold_bn: bool = True,
                 use_graphs: Optional[bool] = None):
        super().__init__()
        from .determinism import maybe_fix_random
        maybe_fix_random()                     # RMEM_DETERMINISTIC=1: the reference's --fix_random (tools/eval.py:21-37)
        self.use_graphs = use_graphs
        self.cfg = aot_model.cfg
        self.AOT = aot_model
        if next(aot_model.parameters()).is_cuda:
            from . import hip
            hip.set_host_wait(next(aot_model.parameters()).device.index or 0)     # (RMEM_BLOCKING_WAIT=1 only: opt-in, hip.set_host_wait)
        if fold_bn and hasattr(aot_model, "optimize_for_inference") and \
                next(aot_model.parameters()).is_cuda:
            aot_model.optimize_for_inference(True)     # weights must already be loaded / on device
        if max_aot_obj_num is None or max_aot_obj_num > aot_model.max_obj_num:
            self.max_aot_obj_num = aot_model.max_obj_num
        else:
            self.max_aot_obj_num = max_aot_obj_num
        self.gpu_id = gpu_id
        self.long_term_mem_gap = long_term_mem_gap
        self.short_term_mem_skip = short_term_mem_skip
        self.nsplit = nsplit
        self.aot_engines: List[DeAOTEngine] = []
        self._pool: List[DeAOTEngine] = []     # engines (and their HBM buffers) are reused across clips
        self._bat = None                       # the BatchedDeAOTEngine whose slots are the sub-engines (> 10 objects), or None
        self._bat_pool: dict = {}              # number of sub-engines -> BatchedDeAOTEngine, reused across clips
        self._adopt_steps = None               # adopt_frame_steps(): frame counters of the sub-engines the next reference frame creates
        self.restart_engine()

    def restart_engine(self):                                   # aot_engine.py:598-602
        for e in self.aot_engines:
            e.restart_engine()
        singles = [e for e in self.aot_engines if isinstance(e, DeAOTEngine)]
        self._pool = singles + [e for e in self._pool if e not in singles]
        self.aot_engines = []
        self._bat = None
        self.obj_nums = None

    def _batched_ok(self, n: int) -> bool:
        """Sub-engines as slots of one batched engine: DeAOT block on the GPU, product sub-engines (tests substitute
        theirs through _new_engine), not switched off."""
        return (n > 1 and self.cfg.MODEL_VOS == "deaot" and next(self.AOT.parameters()).is_cuda
                and type(self)._new_engine is DeAOTInferEngine._new_engine
                and os.environ.get("RMEM_MULTI_ENGINE", "batched") != "serial")

    def _add_reference_batched(self, img, mask, aot_num: int, frame_step: int):
        from .batched import BatchedDeAOTEngine
        # every engine of the reference keeps its own frame counter: those that exist carry on, new ones start at 0
        counters = [int(e.frame_step) for e in self.aot_engines][:aot_num]
        adopt, self._adopt_steps = (self._adopt_steps or []), None
        counters += [int(adopt[i]) if i < len(adopt) else 0 for i in range(len(counters), aot_num)]
        bat = self._bat
        if bat is None or bat.B != aot_num:
            singles = [e for e in self.aot_engines if isinstance(e, DeAOTEngine)]      # (a clip that grows past 10 objects)
            for e in singles:
                e.restart_engine()
            self._pool = singles + [e for e in self._pool if e not in singles]
            bat = self._bat_pool.get(aot_num)
            if bat is None:
                bat = self._bat_pool[aot_num] = BatchedDeAOTEngine(
                    self.AOT, aot_num, gpu_id=self.gpu_id, long_term_mem_gap=self.long_term_mem_gap, nsplit=self.nsplit,
                    use_graphs=self.use_graphs)
            bat.restart_engine()
        bat.long_term_mem_gap = self.long_term_mem_gap
        bat.frame_steps = counters
        self._bat = bat
        self.aot_engines = [_SubEngineView(bat, i) for i in range(aot_num)]
        masks = torch.cat([m.reshape(1, 1, *m.shape[-2:]) for m in self.separate_mask(mask)])
        bat.add_reference_frame(img, masks, obj_nums=[self.max_aot_obj_num] * aot_num, frame_step=frame_step)
        self.update_size()

    def adopt_frame_steps(self, steps):
        """The sub-engines the NEXT add_reference_frame creates take these frame counters instead of 0 (one entry per
        sub-engine, in order).  A wrapper that takes over a clip mid-way (rmem_amd.driver: test-time augmentation handed
        from the batched-augmentation engine to per-augmentation engines when a new label exceeds ten objects) stands in
        for engines that ran since frame 0: in the reference those keep counting (long_memories_indexes and the gap
        schedule follow the engine's own counter, aot_engine.py:322-323, 338-343), only sub-engines that appear with the
        new label start at 0."""
        self._adopt_steps = [int(v) for v in steps]

    def _new_engine(self) -> DeAOTEngine:
        """One sub-engine (<= max_aot_obj_num objects); tests substitute engines that run the encoder and
        the decoder on the CPU around the HIP LSTT (tests/sandwich.py)."""
        return DeAOTEngine(self.AOT, self.gpu_id, self.long_term_mem_gap, self.short_term_mem_skip, self.nsplit,
                           self.use_graphs)

    def separate_mask(self, mask):                              # aot_engine.py:604-628
        if mask is None:
            return [None] * len(self.aot_engines)
        if len(self.aot_engines) == 1:
            return [mask]
        if mask.dim() == 3 or mask.shape[0] == 1:
            out = []
            for idx in range(len(self.aot_engines)):
                start_id = idx * self.max_aot_obj_num + 1
                end_id = (idx + 1) * self.max_aot_obj_num
                fg = ((mask >= start_id) & (mask <= end_id)).float()
                out.append((fg * mask - start_id + 1) * fg)
            return out
        raise NotImplementedError("probability masks are a training-only input")

    def soft_logit_aggregation(self, all_logits):               # aot_engine.py:650-673
        if len(all_logits) == 1:
            return all_logits[0]
        fg_probs, bg_probs = [], []
        for logit in all_logits:
            prob = torch.softmax(logit, dim=1)
            bg_probs.append(prob[:, 0:1])
            fg_probs.append(prob[:, 1:1 + self.max_aot_obj_num])
        bg_prob = torch.prod(torch.cat(bg_probs, dim=1), dim=1, keepdim=True)
        merged = torch.cat([bg_prob] + fg_probs, dim=1).clamp(1e-5, 1 - 1e-5)
        return torch.logit(merged)

    def add_reference_frame(self, img, mask, obj_nums, frame_step=-1):     # deaot_engine.py:30-56
        if isinstance(obj_nums, list):
            obj_nums = obj_nums[0]
        self.obj_nums = obj_nums
        aot_num = max(int(np.ceil(obj_nums / self.max_aot_obj_num)), 1)
        aot_num = max(aot_num, len(self.aot_engines))           # (engines are never dropped inside a clip: `while aot_num > len`)
        if self._batched_ok(aot_num):
            return self._add_reference_batched(img, mask, aot_num, frame_step)
        adopt, self._adopt_steps = (self._adopt_steps or []), None
        while aot_num > len(self.aot_engines):
            if self._pool:
                eng = self._pool.pop(0)
                eng.long_term_mem_gap = self.long_term_mem_gap
            else:
                eng = self._new_engine()
            eng.eval()
            if len(self.aot_engines) < len(adopt):              # (adopt_frame_steps: this sub-engine stands in for one that ran the clip so far)
                eng.frame_step = int(adopt[len(self.aot_engines)])
            self.aot_engines.append(eng)
        img_embs = self.AOT.encode_image(img) if len(self.aot_engines) > 1 else None    # shared encoder pass
        for eng, m in zip(self.aot_engines, self.separate_mask(mask)):
            eng.add_reference_frame(img, m, obj_nums=[self.max_aot_obj_num], frame_step=frame_step,
                                    img_embs=img_embs)
        self.update_size()

    @torch.no_grad()
    def match_propogate_one_frame(self, img=None, mask=None, output_size=None, next_img=None):
        """aot_engine.py:704-712.  With several sub-engines (> 10 objects) the image is encoded
        once and the features are shared (the reference re-encodes per sub-engine: img_embs stays
        None at :705-709).  `next_img`: see DeAOTEngine.match_propogate_one_frame."""
        if len(self.aot_engines) == 1:
            return self.aot_engines[0].match_propogate_one_frame(img, mask=mask, output_size=output_size,
                                                                 next_img=next_img)
        if self._bat is not None:
            nxt = next_img[0] if isinstance(next_img, (list, tuple)) and next_img else next_img
            lg = self._bat.match_propogate_one_frame(img, output_size=output_size,
                                                     next_imgs=nxt if torch.is_tensor(nxt) else None)
            return self.soft_logit_aggregation([lg[i:i + 1] for i in range(self._bat.B)])
        img_embs = self.AOT.encode_image(img)
        all_logits = [e.match_propogate_one_frame(img, img_embs=img_embs, mask=mask, output_size=output_size)
                      for e in self.aot_engines]
        return self.soft_logit_aggregation(all_logits)

    def update_memory(self, curr_mask):                         # aot_engine.py:714-720
        if self._bat is not None:
            return self._bat.update_memory(torch.cat([m.reshape(1, 1, *m.shape[-2:])
                                                      for m in self.separate_mask(curr_mask)]))
        for eng, m in zip(self.aot_engines, self.separate_mask(curr_mask)):
            eng.update_short_term_memory(m)

    def update_size(self):                                      # aot_engine.py:722-725
        self.input_size_2d = self.aot_engines[0].input_size_2d
        self.enc_size_2d = self.aot_engines[0].enc_size_2d
        self.enc_hw = self.aot_engines[0].enc_hw


def build_engine(name, phase="eval", **kwargs):
    """engines/__init__.py:5-21 (inference phases only; training is out of scope)."""
    if phase != "eval":
        raise NotImplementedError("only phase='eval' is built (training is out of the hot-path scope)")
    if name in ("deaotengine", "aotengine"):     # one engine class; the LSTT follows cfg.MODEL_VOS
        return DeAOTInferEngine(**kwargs)
    raise NotImplementedError(name)
