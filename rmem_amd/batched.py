"""Several clips per GPU in ONE launch per kernel.

The reference serves one clip per process: its attention asserts batch 1
(/root/reference/aot_plus/networks/layers/transformer.py:641,1190) and the evaluator walks the clips
one after the other (managers/evaluator.py:344-523).  Here B clips of one geometry -- the slots of a
batch -- share the launches of the memory path:

  * each clip keeps its own ``DeAOTLSTT`` state (bank ring, slot map, eviction bookkeeping);
    their packed weights are shared;
  * the host code of a pass is RECORDED once per (clip, slot, bank depth) -- librmem_hip's launch
    recorder (include/rmem_hip.h, csrc/launch.h) turns every entry point into "append the
    argument block" -- and the argument blobs are uploaded (one pinned-memory copy) and replayed
    by ``rmem_launch_recorded``: per kernel ONE launch whose grid covers the clips, each block
    reading its clip's arguments from device memory.  Per clip the arithmetic is, bit for bit, that of a
    single-clip ``DeAOTLSTT`` built with ``clips_per_launch=B`` -- the same kernels' bodies with the same key splits
    and the same four-way split-K of the projections (tests/test_hip_batched.py).  The ONE-clip engine
    (``clips_per_launch=1``) runs the projections on the streaming kernel with a two-way split-K and folds the last
    partials in the GroupNorm: fp32-equal, not bit-equal, so a near-tie pixel may differ between a clip run alone
    and the same clip in a batch -- which is why ``BatchedClipDriver.run_dataset`` sends even a lone clip of its
    frame size through the slot queue (clip hashes must not depend on how clips are grouped over ranks);
  * clips in the SAME state (same pass, same bank depth) share a launch; a clip in another state -- a
    slot that has just taken the next clip of a queue (reference frame, bank still filling), a clip on
    another gap schedule -- gets launches of its own group (``BatchedLSTT._run`` groups by recording
    signature), an idle slot takes part in none.  Equal-length clips in lockstep (config 4 of
    BASELINE.json) are one group throughout;
  * encoder and decoder run through MIOpen at batch B (hipGraphs keyed by the shapes only: nothing
    clip-specific is baked in, the per-clip state lives in the uploaded argument blocks); ONE image may
    be shared by all slots (the sub-engines of a clip with more than 10 objects: encoder at batch 1);
  * the RMem eviction runs on the device per clip (rmem_bank_policy_step; RMEM_HOST_POLICY=1: one
    device-to-host copy per long-term update for all clips and the rule on the host).

``BatchedDeAOTEngine`` mirrors the engine API of engines/aot_engine.py with a leading clip axis:
``add_reference_frame(imgs [B,3,H,W], masks [B,1,H,W], obj_nums=[n_0..n_B-1])``,
``match_propogate_one_frame(imgs, ref_slots=..., idle_slots=...)`` -> logits [B,C,H,W], ``update_memory(masks)``.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import hip
from .lstt import DeAOTLSTT


class _Upload:
    """Stream-ordered host -> device upload of a small byte block through a ring of pinned buffers
    (the device buffer keeps its address: it is baked into recorded argument blocks)."""

    RING = 6

    def __init__(self, nbytes: int, device):
        self.nbytes = int(nbytes)
        self.dev = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
        self.pin = [torch.zeros(self.nbytes, dtype=torch.uint8).pin_memory() for _ in range(self.RING)]
        self.np = [p.numpy() for p in self.pin]
        self.done: List[Optional[torch.cuda.Event]] = [None] * self.RING
        self.k = 0

    def stage(self) -> np.ndarray:
        """The pinned buffer to fill next (waits until the copy that last read it has run)."""
        self.k = (self.k + 1) % self.RING
        if self.done[self.k] is not None:
            hip.wait_event(self.done[self.k])
        return self.np[self.k]

    def send(self):
        self.dev.copy_(self.pin[self.k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.done[self.k] = ev


class BatchedLSTT:
    """B ``DeAOTLSTT`` states of one geometry served by shared launches."""

    def __init__(self, model, h: int, w: int, device, B: int, nsplit: int = 3):
        self.B = int(B)
        first = DeAOTLSTT(model, h, w, device, nsplit, clips_per_launch=self.B)
        self.clips: List[DeAOTLSTT] = [first] + [
            DeAOTLSTT(model, h, w, device, nsplit, clips_per_launch=self.B, weights_from=first)
            for _ in range(self.B - 1)]
        self.dev = first.dev
        self.N, self.h, self.w = first.N, first.h, first.w
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=self.dev)
        self.tgt, self.tgt_id = z(self.B, self.N, 256), z(self.B, self.N, 256)
        self.out = z(self.B, self.N, 512)
        self.w_out = z(self.B, first.Tmax)
        self.fg = z(self.B, self.N)
        self.maps_up = _Upload(self.B * 128, self.dev)
        maps_i32 = self.maps_up.dev.view(torch.int32).view(self.B, 32)
        self.maps_i32 = maps_i32
        # RMem eviction rule on the device, per clip (rmem_bank_policy_step; RMEM_HOST_POLICY=1: on the host, with ONE
        # blocking device-to-host copy per long-term update for all clips): the bank maps then live on the device and a
        # pass publishes only [short-term slot, slot of this frame] per clip
        self.device_policy = os.environ.get("RMEM_HOST_POLICY") != "1"
        self.short_up = _Upload(self.B * 8, self.dev)
        self._short_host = np.zeros((self.B, 2), dtype=np.int32)
        for i, c in enumerate(self.clips):
            c.tgt, c.tgt_id, c.out, c.w_out = self.tgt[i], self.tgt_id[i], self.out[i], self.w_out[i]
            c.maps = maps_i32[i]
            c._batched = True
            c.device_policy = self.device_policy
        self._labels: Dict[tuple, torch.Tensor] = {}
        self._recs: Dict[tuple, hip.Recording] = {}
        self._chan: Dict[tuple, _Upload] = {}
        self._maps_host: Optional[np.ndarray] = None      # what the clips last published (slot maps), all clips
        self.groups_last = 0         # launch groups of the last pass (1 = every active clip in the same state)
        self.launches = 0            # recorded ops issued (each ONE launch for all clips)

    # ------------------------------------------------------------------ recorder plumbing
    def _record(self, kind: str, i: int, key: tuple, fn) -> hip.Recording:
        k = (kind, i) + key
        r = self._recs.get(k)
        if r is None:
            with hip.Recording() as r:
                fn(self.clips[i])
            self._recs[k] = r
        return r

    def _run(self, kind: str, keys: List[tuple], fn, active: Optional[List[bool]] = None):
        """Record (or look up) `fn(clip)` for every active clip, upload the argument blobs, launch.  Clips whose
        recordings share a signature (same kernels, grids, argument-block offsets: same pass at the same bank
        depth) share ONE launch per kernel; a clip in another state (a slot that has just taken the next clip of
        the queue: reference frame, shallower bank) gets launches of its own group.  keys[i] is ignored for an
        inactive clip."""
        idx = [i for i in range(self.B) if active is None or active[i]]
        groups: Dict[tuple, List[int]] = {}
        recs = {}
        for i in idx:
            r = recs[i] = self._record(kind, i, keys[i], fn)
            groups.setdefault((r.signature, len(r.blob)), []).append(i)
        for (sig, n), members in groups.items():
            stride = (n + 255) // 256 * 256
            ch = self._chan.get((kind, n))
            if ch is None:
                ch = self._chan[(kind, n)] = _Upload(self.B * stride, self.dev)
            buf = ch.stage()
            for pos, i in enumerate(members):
                buf[pos * stride:pos * stride + n] = np.frombuffer(recs[i].blob, dtype=np.uint8)
            ch.send()
            recs[members[0]].launch(ch.dev, stride, len(members))
            self.launches += recs[members[0]].count
        self.groups_last = len(groups)

    @staticmethod
    def _per_clip(v, B: int) -> list:
        return list(v) if isinstance(v, (list, tuple)) else [v] * B

    # ------------------------------------------------------------------ passes
    def label_buffer(self, H: int, W: int) -> torch.Tensor:
        """Static uint8 [B,H,W] buffer the ID assignment reads the clips' label maps from."""
        buf = self._labels.get((H, W))
        if buf is None:
            buf = self._labels[(H, W)] = torch.zeros(self.B, H, W, dtype=torch.uint8, device=self.dev)
        return buf

    def clear_memory(self, clip: Optional[int] = None):
        for c in (self.clips if clip is None else [self.clips[clip]]):
            c.clear_memory()

    def assign_identity(self, labels_u8: torch.Tensor, ignore=True, active: Optional[List[bool]] = None):
        """labels_u8: the label_buffer() of its shape (or a tensor copied into it).  `ignore`: one flag or one per
        clip; `active`: the clips that take part (None = all)."""
        B, H, W = labels_u8.shape
        lab = self.label_buffer(H, W)
        if labels_u8.data_ptr() != lab.data_ptr():
            lab.copy_(labels_u8)
        ign = [bool(g) for g in self._per_clip(ignore, self.B)]
        self._run("id", [(H, W, ign[i]) for i in range(self.B)],
                  lambda c: c.assign_identity(lab[self.clips.index(c)], ignore=ign[self.clips.index(c)]), active)

    def forward(self, emb_bnc: torch.Tensor, ref_frame=False, active: Optional[List[bool]] = None) -> torch.Tensor:
        """DualBranchGPM.forward for every (active) clip.  emb_bnc: [B,N,256] fp32 -> [B,N,512].  `ref_frame`: one
        flag or one per clip -- a slot may add the reference frame of its next clip while the others propagate."""
        ref = [bool(r) for r in self._per_clip(ref_frame, self.B)]
        on = [True] * self.B if active is None else [bool(a) for a in active]
        self.tgt.copy_(emb_bnc)
        self.tgt_id.zero_()
        if self.device_policy:             # bank maps on the device: [short, cur] per clip, scattered into the maps
            stage = self.short_up.stage().view(np.int32).reshape(self.B, 2)
            for i, c in enumerate(self.clips):
                if on[i]:
                    c._prepare(ref[i])         # (reference frame: rmem_bank_reset for that clip)
                    self._short_host[i] = c._map_vals[16:18]
            stage[:] = self._short_host        # (inactive clips keep what they published last)
            self.short_up.send()
            self.maps_i32[:, 16:18].copy_(self.short_up.dev.view(torch.int32).view(self.B, 2))
        else:
            stage = self.maps_up.stage().view(np.int32).reshape(self.B, 32)
            if self._maps_host is not None:
                stage[:] = self._maps_host
            for i, c in enumerate(self.clips):
                if on[i]:
                    c._prepare(ref[i])
                    stage[i, :18] = c._map_vals
            self._maps_host = stage.copy()
            self.maps_up.send()
        self._run("fwd", [(c.cur, c._T, ref[i]) if on[i] else () for i, c in enumerate(self.clips)],
                  lambda c: c._forward_device(ref[self.clips.index(c)]), on)
        for i, c in enumerate(self.clips):
            if on[i]:
                c._finish(ref[i])
        return self.out

    def time_read_isolated(self, iters: int = 20) -> float:
        """Mean duration (us) of ONE launch of the fused long-term + windowed read of layer 0 for all
        clips (read2_many_kernel) on the current bank state, with nothing else on the GPU."""
        import ctypes as C

        def issue(c):
            T = len(c.bank)
            c._layer = 0
            A = c._read_args(c.ws_main, 0, T, c.bankK[0], c.bankV[0], c.maps.data_ptr(), c.Qpe, c.bias_pe, c.Ucat0,
                             True, c.ks_long)
            Bq = c._read_args(c.ws_side, 1, 1, c.bankK[0], c.bankV[0], c.maps.data_ptr() + 64,
                              hip.Planes(c.bankK[0].hi[c.cur], c.bankK[0].lo[c.cur]), None, c.Ucat0, False, c.ks_win)
            hip.check(hip.load().rmem_attn_read2(C.byref(A[0]), C.byref(Bq[0]), hip.stream_ptr()), "rmem_attn_read2")
        keys = [(c.cur, len(c.bank)) for c in self.clips]
        torch.cuda.synchronize()
        for _ in range(3):
            self._run("probe_read2", keys, issue)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            self._run("probe_read2", keys, issue)
        e1.record()
        e1.synchronize()
        return 1e3 * e0.elapsed_time(e1) / iters

    def update_short_memories(self, update_long, active: Optional[List[bool]] = None, frame_index=0):
        """`update_long`: one flag or one per clip (clips of different lengths follow different gap schedules);
        `frame_index`: the frame step(s) the engine appends to long_memories_indexes."""
        upd = [bool(u) for u in self._per_clip(update_long, self.B)]
        fi = [int(f) for f in self._per_clip(frame_index, self.B)]
        self._run("upd", [(c.cur,) for c in self.clips], lambda c: c._update_device(upd[self.clips.index(c)]), active)
        for i, c in enumerate(self.clips):
            if active is None or active[i]:
                c._update_host(upd[i], fi[i])

    def resolve_policy(self, block: bool = True) -> bool:
        """Host views (clip.bank, the indexes lists handed to restrict_long_memories) up to date with the device-side
        evictions of every clip."""
        return all([c.resolve_policy(block) for c in self.clips])

    def restrict_long_memories(self, indexes: List[List[int]], fg_bn: torch.Tensor, active: Optional[List[bool]] = None,
                               wait: bool = False):
        """restrict_long_memories (transformer.py:880-991) for every (active) clip: one reduce launch per group of
        clips with the same bank depth, then per clip the EMA + UCB rule on the device (rmem_bank_policy_step:
        nothing is read back here; `indexes[i]` and the clip's bank catch up in resolve_policy()) -- or, with
        RMEM_HOST_POLICY=1, ONE device-to-host copy and the rule on the host.  Returns the dropped position (or
        None) per clip when the rule ran on the host or `wait` is set, else None for every clip."""
        on = [True] * self.B if active is None else [bool(a) for a in active]
        if not any(on):
            return [None] * self.B
        self.fg.copy_(fg_bn)
        self._run("mass", [(c.mass_T,) for c in self.clips],
                  lambda c: c._mass_reduce_device(self.fg[self.clips.index(c)]), on)
        if self.device_policy:
            for i, c in enumerate(self.clips):
                if on[i]:
                    c._policy_device(indexes[i])
            if not wait:
                return [None] * self.B
            drops = []
            for i, c in enumerate(self.clips):
                if on[i]:
                    c.resolve_policy(block=True)
                drops.append(c.last_policy["drop"] if on[i] and c.last_policy["drop"] >= 0 else None)
            return drops
        T = max(c.mass_T for i, c in enumerate(self.clips) if on[i])
        w = self.w_out[:, :T].cpu().numpy().astype(np.float32)
        return [c._restrict_host(indexes[i], w[i, :c.mass_T]) if on[i] else None for i, c in enumerate(self.clips)]


class BatchedDeAOTEngine:
    """B clips (one engine each in the reference, engines/aot_engine.py:18-568) in lockstep."""

    def __init__(self, aot_model, B: int, gpu_id: int = 0, long_term_mem_gap: int = 9999, nsplit: int = 3,
                 fold_bn: bool = True, use_graphs: Optional[bool] = None):
        if aot_model.cfg.MODEL_VOS != "deaot":
            raise NotImplementedError("batched clips are built for the DeAOT + RMem path")
        self.AOT = aot_model
        self.cfg = aot_model.cfg
        self.B = int(B)
        self.gpu_id = gpu_id
        self.long_term_mem_gap = long_term_mem_gap
        self.nsplit = nsplit
        self.align_corners = self.cfg.MODEL_ALIGN_CORNERS
        if use_graphs is None:
            use_graphs = os.environ.get("RMEM_NO_GRAPHS") is None
        self.use_graphs = bool(use_graphs)
        if next(aot_model.parameters()).is_cuda:
            hip.set_host_wait(next(aot_model.parameters()).device.index or 0)     # (RMEM_BLOCKING_WAIT=1 only: opt-in, hip.set_host_wait)
        if fold_bn and next(aot_model.parameters()).is_cuda:
            aot_model.optimize_for_inference(True)
        self.lstt: Optional[BatchedLSTT] = None
        self._lstt_wv = 0                    # weights version (rmem_amd.checkpoint.load_network) the packed LSTT was built from
        self._eg, self._dg = {}, {}
        self._par = 0                        # encoder feature copy (of two) that holds the current frame
        self._pending = None                 # (images, copy, done event) of the announced next frame
        self._enc_stream = None
        self.restart_engine()

    def restart_engine(self):                                   # aot_engine.py:533-563
        self.frame_steps = [0] * self.B           # per slot: a slot that takes the next clip of a queue starts over
        self.last_mem_steps = [-1] * self.B
        self.slot_gaps: List[Optional[int]] = [None] * self.B      # per-slot gap (None: long_term_mem_gap)
        self.active = [True] * self.B             # slots that hold a clip (an idle slot takes part in no launch of the memory path)
        self._ref_now: set = set()                # slots whose current frame is a reference frame (no memory update follows)
        self.obj_nums = None
        self.input_size_2d = self.enc_size_2d = self.enc_hw = None
        self.long_memories_indexes: List[List[int]] = [[] for _ in range(self.B)]
        self.pred_id_logits = None
        self._drop_pending()
        if self._stale_weights():
            # load_network() on a model this engine already holds: the packed LSTT / ID-bank planes and the
            # captured encoder / decoder graphs (which replay the OLD folded tensors) are stale -- drop them,
            # as DeAOTEngine.restart_engine does; the next reference frame re-packs and re-captures
            if self._eg or self._dg:
                torch.cuda.synchronize()
            self.lstt, self._eg, self._dg = None, {}, {}
        if self.lstt is not None:
            self.lstt.clear_memory()
            for c in self.lstt.clips:
                c.ref_frame_index = 0
        self._cur_tokens = None               # encoder tokens of the frame the last call propagated (add_reference_slots)

    @property
    def long_memories_indexes(self) -> List[List[int]]:
        """Frame indexes of the bank slots, per clip (aot_engine.py:322, 346).  With the eviction rule on the device the
        host's copy may lag by one decision per clip; reading it waits for those decisions."""
        if self.__dict__.get("lstt") is not None:
            self.lstt.resolve_policy(block=True)
        return self._lmi

    @long_memories_indexes.setter
    def long_memories_indexes(self, v):
        self._lmi = v

    # lockstep callers (every slot on the same frame) read and write one counter
    @property
    def frame_step(self) -> int:
        return self.frame_steps[0]

    @frame_step.setter
    def frame_step(self, v: int):
        self.frame_steps = [int(v)] * self.B

    @property
    def last_mem_step(self) -> int:
        return self.last_mem_steps[0]

    @last_mem_step.setter
    def last_mem_step(self, v: int):
        self.last_mem_steps = [int(v)] * self.B

    def restart_slot(self, i: int, gap: Optional[int] = None):
        """restart_engine (aot_engine.py:533-563) for ONE slot: the reference's worker takes the next clip of the
        queue when its clip ends (managers/evaluator.py:287-295); here the slot does, while the others carry on."""
        self.frame_steps[i], self.last_mem_steps[i] = 0, -1
        self.slot_gaps[i] = gap
        self._lmi[i] = []
        self.active[i] = True
        self.lstt.clear_memory(i)
        self.lstt.clips[i].ref_frame_index = 0

    def _stale_weights(self) -> bool:
        return self._lstt_wv != self.AOT.__dict__.get("_weights_version", 0)

    def update_size(self, input_size, enc_size):                # aot_engine.py:565-568
        self.input_size_2d = tuple(int(v) for v in input_size)
        self.enc_size_2d = tuple(int(v) for v in enc_size)
        self.enc_hw = self.enc_size_2d[0] * self.enc_size_2d[1]
        if self.lstt is None or (self.lstt.h, self.lstt.w) != self.enc_size_2d or self._stale_weights():
            replaced = self.lstt is not None
            self._lstt_wv = self.AOT.__dict__.get("_weights_version", 0)
            dev = next(self.AOT.parameters()).device
            self.lstt = BatchedLSTT(self.AOT, self.enc_size_2d[0], self.enc_size_2d[1], dev, self.B, self.nsplit)
            if replaced:       # geometry / weights changed under live graphs (a first LSTT invalidates nothing:
                self._drop_pending()      # encoder and decoder graphs do not depend on it)
                torch.cuda.synchronize()
                self._eg, self._dg = {}, {}

    # ------------------------------------------------------------------ encoder / decoder at batch B
    def _encoder_graph(self, imgs: torch.Tensor, par: int):
        key = (tuple(imgs.shape), par)
        ent = self._eg.get(key)
        if ent is None:
            g_img = torch.zeros_like(imgs)
            self.AOT.encode_image(g_img)                  # MIOpen picks its solvers outside the capture
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                enc = self.AOT.encode_image(g_img)
            ent = self._eg[key] = (g, g_img, enc)
        return ent

    def _drop_pending(self):
        p, self._pending = getattr(self, "_pending", None), None
        if p is not None:
            torch.cuda.current_stream().wait_event(p[2])

    def _encode(self, imgs: torch.Tensor):
        """Encoder pyramid of the B frames.  With hipGraphs there are two feature copies: the pass of
        an announced next frame (`_prefetch`) runs on a second stream into the other copy while
        this frame's LSTT and decoder read theirs."""
        if not (self.use_graphs and imgs.is_cuda):
            return self.AOT.encode_image(imgs)
        p, self._pending = self._pending, None
        if p is not None:
            torch.cuda.current_stream().wait_event(p[2])
            if p[0] is imgs:
                self._par = p[1]
                return self._eg[(tuple(imgs.shape), self._par)][2]
        ent = self._encoder_graph(imgs, self._par)
        ent[1].copy_(imgs)
        ent[0].replay()
        return ent[2]

    def _prefetch(self, next_imgs):
        if next_imgs is None or not (self.use_graphs and next_imgs.is_cuda) or self._pending is not None:
            return
        par = 1 - self._par
        ent = self._encoder_graph(next_imgs, par)
        if self._enc_stream is None:
            from .streams import concurrent_stream
            self._enc_stream = concurrent_stream(next_imgs.device)
        after, done = torch.cuda.Event(), torch.cuda.Event()
        after.record(torch.cuda.current_stream())     # the previous reader of that copy is queued before this point
        with torch.cuda.stream(self._enc_stream):
            self._enc_stream.wait_event(after)
            ent[1].copy_(next_imgs, non_blocking=True)
            next_imgs.record_stream(self._enc_stream)
            ent[0].replay()
            done.record(self._enc_stream)
        self._pending = (next_imgs, par, done)

    def _decode_eager(self, enc, osz):
        h, w = self.enc_size_2d
        emb = self.lstt.out.view(self.B, h, w, 512).permute(0, 3, 1, 2)
        if enc[-1].shape[0] == 1 and self.B > 1:
            # ONE image shared by all slots (the sub-engines of a many-object clip): the encoder ran at batch 1; the
            # decoder's inputs get the clip axis here (views; the skip-adapter outputs are read by a HIP kernel: copies)
            from .model import FeatureList
            shared = enc
            enc = FeatureList([e.expand(self.B, -1, -1, -1) for e in shared])
            if getattr(shared, "adapters", None) is not None:
                enc.adapters = [a.expand(self.B, -1, -1, -1).contiguous() for a in shared.adapters]
        logits = self.AOT.decoder([enc[-1], emb], enc)
        for b, obj_num in enumerate(self.obj_nums):
            logits[b, (obj_num + 1):] = -1e10
        up = logits if osz is None else F.interpolate(logits, size=osz, mode="bilinear",
                                                      align_corners=self.align_corners)
        return logits, up

    def _decode(self, enc, output_size, graph_ok: bool):
        osz = tuple(int(v) for v in output_size) if output_size is not None else None
        if not (self.use_graphs and graph_ok):
            return self._decode_eager(enc, osz)
        key = (osz, id(enc), tuple(self.obj_nums))
        ent = self._dg.get(key)
        if ent is None and len(self._dg) >= 8:        # a caller that keeps changing the output size: bounded cache
            torch.cuda.synchronize()
            self._dg = {}
        if ent is None:
            self._decode_eager(enc, osz)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                logits, up = self._decode_eager(enc, osz)
            ent = self._dg[key] = (g, logits, up, enc)
        ent[0].replay()
        return ent[1], ent[2]

    def _labels_u8(self, masks: torch.Tensor) -> torch.Tensor:
        m = masks
        if m.dim() == 4:
            if m.shape[1] != 1:
                raise NotImplementedError("probability masks are a training-only input (aot_engine.py:333-334)")
            m = m[:, 0]
        if m.dim() != 3 or m.shape[0] != self.B:
            raise ValueError(f"masks must be [B,1,H,W] or [B,H,W] with B = {self.B}")
        return m.to(torch.uint8).contiguous()

    # ------------------------------------------------------------------ engine API
    @torch.no_grad()
    def add_reference_frame(self, imgs, masks, obj_nums, frame_step: int = -1):
        """aot_engine.py:241-325 for B clips.  obj_nums: one object count per clip."""
        if len(obj_nums) != self.B or imgs.shape[0] not in (1, self.B):
            raise ValueError("one mask and object count per clip; one image per clip, or ONE image shared by all slots "
                             "(the sub-engines of a clip with more than 10 objects, DeAOTInferEngine)")
        self.obj_nums = [int(n) for n in obj_nums]
        if any(n > self.AOT.max_obj_num for n in self.obj_nums):
            raise NotImplementedError(f"more than {self.AOT.max_obj_num} objects per clip: use DeAOTInferEngine (one sub-engine "
                                      "per 10 objects, engines/aot_engine.py:675-702)")
        self._drop_pending()
        if self._stale_weights() and (self._eg or self._dg):    # load_network() since the graphs were captured
            torch.cuda.synchronize()
            self._eg, self._dg = {}, {}
        enc = self._encode(imgs)
        if self.input_size_2d is None or self._stale_weights():
            self.update_size(imgs.shape[2:], enc[-1].shape[2:])
        self.active, self._ref_now = [True] * self.B, set()
        # no ignore channel on reference frames (aot_engine.py:304 -> :209-213)
        self.lstt.assign_identity(self._labels_u8(masks), ignore=False)
        self._cur_tokens = None
        self.lstt.forward(enc[-1].flatten(2).transpose(1, 2), ref_frame=True)
        # per slot, as every engine of the reference does for itself (aot_engine.py:241-325: last_mem_step := the
        # frame_step argument, or the engine's own counter; long_memories_indexes := [its own counter])
        self.last_mem_steps = list(self.frame_steps) if frame_step == -1 else [int(frame_step)] * self.B
        self.long_memories_indexes = [[fs] for fs in self.frame_steps]
        self.pred_id_logits, _ = self._decode(enc, None, imgs.is_cuda)

    @torch.no_grad()
    def match_propogate_one_frame(self, imgs, output_size=None, next_imgs=None, ref_slots: Optional[Dict] = None,
                                  idle_slots=()):
        """aot_engine.py:398-436 for B clips -> logits [B,C,H,W].  `next_imgs` (extension, optional):
        the tensor that will be passed to the NEXT call; its encoder pass then runs on a second
        stream beside this frame's LSTT and decoder (the encoder does not depend on the memory).

        `ref_slots` (extension): {slot: (label map [H,W] / [1,1,H,W] at the network size, object count, gap or None)}
        -- imgs[slot] is the reference frame of the NEXT clip of that slot: the slot starts over
        (restart_slot + add_reference_frame, aot_engine.py:241-325) inside this step while the other slots
        propagate; its row of the returned logits means nothing and update_memory() skips it.  `idle_slots`: slots
        without a clip (queue empty): their images still pass the encoder / decoder at batch B, the memory path
        skips them."""
        ref_slots = ref_slots or {}
        enc = self._encode(imgs)
        self._prefetch(next_imgs)
        if self.input_size_2d is None or self._stale_weights():
            self.update_size(imgs.shape[2:], enc[-1].shape[2:])
        for i in idle_slots:
            self.active[i] = False
        if ref_slots:
            if self.obj_nums is None:
                self.obj_nums = [int(self.AOT.max_obj_num)] * self.B
            lab = self.lstt.label_buffer(*self.input_size_2d)
            for i, (m, n, gap) in ref_slots.items():
                if int(n) > self.AOT.max_obj_num:
                    raise NotImplementedError(f"more than {self.AOT.max_obj_num} objects per clip: use DeAOTInferEngine")
                if int(n) != self.obj_nums[i]:
                    self.obj_nums = self.obj_nums[:i] + [int(n)] + self.obj_nums[i + 1:]
                self.restart_slot(i, gap)
                lab[i].copy_(m.reshape(lab.shape[1:]).to(torch.uint8))
                self.last_mem_steps[i] = 0
                self._lmi[i] = [0]
            self.lstt.assign_identity(lab, ignore=False, active=[i in ref_slots for i in range(self.B)])
        self._ref_now = set(ref_slots)
        for i in range(self.B):
            if self.active[i] and i not in ref_slots:
                self.frame_steps[i] += 1
        self._cur_tokens = enc[-1].flatten(2).transpose(1, 2)
        self.lstt.forward(self._cur_tokens, ref_frame=[i in ref_slots for i in range(self.B)], active=self.active)
        self.pred_id_logits, up = self._decode(enc, output_size, imgs.is_cuda)
        return up

    @torch.no_grad()
    def add_reference_slots(self, refs: Dict[int, tuple]):
        """A mid-clip reference frame for single slots -- the evaluator's answer to a frame that brings a label with NEW
        objects (managers/evaluator.py:484-508: propagate, paste the new ids over the prediction, then
        add_reference_frame(img, merged label) instead of update_memory).  refs: {slot: (merged label map at the network
        size, object count)}; the frame is the one match_propogate_one_frame() has just propagated for that slot, so its
        encoder features are at hand (the reference encodes the image a second time, aot_engine.py:262-270: same values).
        Per slot what DeAOTEngine.add_reference_frame does mid-clip (aot_engine.py:241-325): ID assignment without the
        ignore channel, a reference-mode pass of the LSTT -- the slot's bank restarts with this frame --, last_mem_step and
        long_memories_indexes := the slot's frame counter.  The other slots take no part; update_memory() of this step
        skips the slots named here.  (The reference-mode logits are not decoded: nothing reads them before the next
        frame's.)"""
        if not refs:
            return
        if self._cur_tokens is None:
            raise RuntimeError("add_reference_slots follows match_propogate_one_frame of the same frame")
        lab = self.lstt.label_buffer(*self.input_size_2d)
        on = [i in refs for i in range(self.B)]
        for i, (m, n) in refs.items():
            if not self.active[i] or i in self._ref_now:
                raise ValueError(f"slot {i} holds no propagated frame in this step")
            if int(n) > self.AOT.max_obj_num:
                raise NotImplementedError(f"more than {self.AOT.max_obj_num} objects per clip: use DeAOTInferEngine")
            if m.data_ptr() != lab[i].data_ptr():
                lab[i].copy_(m.reshape(lab.shape[1:]).to(torch.uint8))
            self.obj_nums = self.obj_nums[:i] + [int(n)] + self.obj_nums[i + 1:]
            self.lstt.clips[i].ref_frame_index = self.frame_steps[i]
        self.lstt.assign_identity(lab, ignore=False, active=on)
        self.lstt.forward(self._cur_tokens, ref_frame=on, active=on)
        for i in refs:
            self.last_mem_steps[i] = self.frame_steps[i]
            self._lmi[i] = [self.frame_steps[i]]
        self._ref_now |= set(refs)

    @torch.no_grad()
    def update_memory(self, masks):
        """update_short_term_memory (aot_engine.py:327-369) for B clips; the long-term update and the RMem eviction
        follow each slot's gap schedule (one shared schedule for clips in lockstep).  Slots that added a reference
        frame in this step, and idle slots, are skipped."""
        on = [self.active[i] and i not in self._ref_now for i in range(self.B)]
        if not any(on):
            return
        upd = [False] * self.B
        if not self.cfg.NO_LONG_MEMORY:
            for i in range(self.B):
                gap = self.slot_gaps[i] if self.slot_gaps[i] is not None else self.long_term_mem_gap
                if on[i] and self.frame_steps[i] - self.last_mem_steps[i] >= gap:
                    upd[i] = True
                    self.last_mem_steps[i] = self.frame_steps[i]
        self.lstt.assign_identity(self._labels_u8(masks), ignore=True, active=on)
        # (a clip that appends to its bank first waits for its previous eviction -- `gap` frames ago, long since there)
        self.lstt.update_short_memories(upd, active=on, frame_index=self.frame_steps)
        if any(upd):
            for i, idx in enumerate(self._lmi):
                if upd[i]:
                    idx.append(self.frame_steps[i])
            lg = F.interpolate(self.pred_id_logits, size=self.enc_size_2d, mode="bilinear", align_corners=True)
            fg = (1 - torch.softmax(lg, dim=1)[:, 0]).reshape(self.B, -1).contiguous()
            self.lstt.restrict_long_memories(self._lmi, fg, active=upd)
