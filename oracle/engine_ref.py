"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the inference
engine state machine (engines/aot_engine.py:241-465,533-568,571-725 and
engines/deaot_engine.py:20-56 of /root/reference/aot_plus) on top of
``oracle.lstt_ref``.  Encoder / decoder are passed in as plain PyTorch modules
(they are outside the hot path and run through PyTorch unchanged).

Parity status: pinned against the imported reference (tests/golden + direct
comparison in tests/test_oracle_vs_reference.py); see oracle/lstt_ref.py.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn.functional as F

from . import lstt_ref as R


class OracleDeAOTEngine:
    """Single-engine (<= 10 objects) DeAOT + RMem inference engine on CPU, fp32."""

    def __init__(self, model, long_term_mem_gap: int = 5):
        self.model = model
        self.cfg = model.cfg
        self.long_term_mem_gap = long_term_mem_gap
        self.sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
        self.lstt = R.DeAOTOracle(self.sd, self.cfg.MODEL_LSTT_NUM)
        self.trace = None
        self.restart_engine()

    def restart_engine(self):                                   # aot_engine.py:533-563
        self.frame_step = 0
        self.last_mem_step = -1
        self.long_memories_indexes: List[int] = []
        self.input_size_2d = None
        self.enc_size_2d = None
        self.enc_hw = None
        self.pred_id_logits = None
        self.lstt.clear_memory()
        self.policy_log = []

    def eval(self):
        return self

    def _decode(self, enc, emb, output_size):                   # aot_engine.py:438-465
        logits = self.model.decode_id_logits(emb, enc)
        # obj_nums is [max_obj] for every sub-engine (aot_engine.py:697), so the
        # "remove unused identities" slice (:451-453) is empty.
        self.pred_id_logits = logits
        if output_size is not None:
            logits = F.interpolate(logits, size=output_size, mode="bilinear",
                                   align_corners=self.cfg.MODEL_ALIGN_CORNERS)
        return logits

    @torch.no_grad()
    def add_reference_frame(self, img, mask, obj_nums=None, frame_step=-1):   # :241-325
        if frame_step == -1:
            frame_step = self.frame_step
        enc = self.model.encode_image(img)
        if self.input_size_2d is None:
            self.input_size_2d = tuple(img.shape[2:])
            self.enc_size_2d = tuple(enc[-1].shape[2:])
            self.enc_hw = self.enc_size_2d[0] * self.enc_size_2d[1]
        h, w = self.enc_size_2d
        # the reference frame's assign_identity gets no ignore mask (aot_engine.py:304, :209-213)
        id_emb = R.id_assign(mask.float(), self.sd, self.cfg.MODEL_MAX_OBJ_NUM, use_ignore=False)
        emb = enc[-1][0].permute(1, 2, 0).reshape(h * w, -1)    # bchw_2_lbc (utils/tensor.py:3-6)
        out = self.lstt.forward(emb, h, w, curr_id_emb=id_emb, trace=self.trace)
        self.last_mem_step = frame_step
        self.lstt.init_memory()
        self.long_memories_indexes.append(self.frame_step)
        self._decode(enc, out, None)

    @torch.no_grad()
    def match_propogate_one_frame(self, img, mask=None, output_size=None):    # :398-436
        self.frame_step += 1
        enc = self.model.encode_image(img)
        h, w = self.enc_size_2d
        emb = enc[-1][0].permute(1, 2, 0).reshape(h * w, -1)
        out = self.lstt.forward(emb, h, w, curr_id_emb=None, trace=self.trace)
        self.last_lstt_out = out
        return self._decode(enc, out, output_size)

    @torch.no_grad()
    def update_memory(self, curr_mask):                         # :327-369, :714-720
        h, w = self.enc_size_2d
        id_emb = R.id_assign(curr_mask.float(), self.sd, self.cfg.MODEL_MAX_OBJ_NUM)
        update_long = False
        if (not self.cfg.NO_LONG_MEMORY) and \
                self.frame_step - self.last_mem_step >= self.long_term_mem_gap:
            update_long = True
            self.last_mem_step = self.frame_step
        self.lstt.update_short_memories(id_emb, update_long)
        if update_long:
            self.long_memories_indexes.append(self.frame_step)
            fg = R.foreground_proba(self.pred_id_logits, h, w)
            log = {"frame": self.frame_step, "indexes_before": list(self.long_memories_indexes)}
            self.lstt.restrict_long_memories(self.cfg.FORMER_MEM_LEN, self.cfg.LATTER_MEM_LEN,
                                             self.long_memories_indexes, fg, log)
            log["indexes_after"] = list(self.long_memories_indexes)
            self.policy_log.append(log)


class OracleAOTEngine(OracleDeAOTEngine):
    """Same state machine for the AOT model (models/aot.py): sine positional embedding once
    per clip (aot_engine.py:289-292), ID embedding without LayerNorm, the decoder consumes the
    encoder embedding plus all three normed LSTT outputs, and the AOT restriction returns
    early while the bank is within its cap (transformer.py:332-334)."""

    def __init__(self, model, long_term_mem_gap: int = 5):
        from . import aot_ref as A
        self.A = A
        super().__init__(model, long_term_mem_gap)
        self.lstt = A.AOTOracle(self.sd, self.cfg.MODEL_LSTT_NUM)
        self.pos = None

    def restart_engine(self):
        super().restart_engine()
        self.pos = None

    def _tokens(self, enc):
        h, w = self.enc_size_2d
        return enc[-1][0].permute(1, 2, 0).reshape(h * w, -1)

    @torch.no_grad()
    def add_reference_frame(self, img, mask, obj_nums=None, frame_step=-1):
        if frame_step == -1:
            frame_step = self.frame_step
        enc = self.model.encode_image(img)
        if self.input_size_2d is None:
            self.input_size_2d = tuple(img.shape[2:])
            self.enc_size_2d = tuple(enc[-1].shape[2:])
            self.enc_hw = self.enc_size_2d[0] * self.enc_size_2d[1]
        h, w = self.enc_size_2d
        if self.pos is None:
            self.pos = self.A.sine_pos_emb(h, w)
        id_emb = self.A.aot_id_assign(mask.float(), self.sd, self.cfg.MODEL_MAX_OBJ_NUM, use_ignore=False)
        outs = self.lstt.forward(self._tokens(enc), h, w, self.pos, curr_id_emb=id_emb, trace=self.trace)
        self.last_mem_step = frame_step
        self.lstt.init_memory()
        self.long_memories_indexes.append(self.frame_step)
        self._decode(enc, outs, None)

    @torch.no_grad()
    def match_propogate_one_frame(self, img, mask=None, output_size=None):
        self.frame_step += 1
        enc = self.model.encode_image(img)
        h, w = self.enc_size_2d
        outs = self.lstt.forward(self._tokens(enc), h, w, self.pos, trace=self.trace)
        self.last_lstt_out = outs
        return self._decode(enc, outs, output_size)

    @torch.no_grad()
    def update_memory(self, curr_mask):
        h, w = self.enc_size_2d
        id_emb = self.A.aot_id_assign(curr_mask.float(), self.sd, self.cfg.MODEL_MAX_OBJ_NUM)
        update_long = False
        if (not self.cfg.NO_LONG_MEMORY) and \
                self.frame_step - self.last_mem_step >= self.long_term_mem_gap:
            update_long = True
            self.last_mem_step = self.frame_step
        self.lstt.update_short_memories(id_emb, update_long)
        if update_long:
            self.long_memories_indexes.append(self.frame_step)
            fg = R.foreground_proba(self.pred_id_logits, h, w)
            self.lstt.restrict_long_memories(self.cfg.FORMER_MEM_LEN, self.cfg.LATTER_MEM_LEN,
                                             self.long_memories_indexes, fg)


class OracleDeAOTInferEngine:
    """AOTInferEngine / DeAOTInferEngine for more than MODEL_MAX_OBJ_NUM objects
    (engines/aot_engine.py:571-725, deaot_engine.py:20-56): one OracleDeAOTEngine per group of
    `max_obj` object ids, each with its OWN memory state, masks separated per engine
    (aot_engine.py:604-628) and decoder logits merged by soft aggregation (:650-673).

    The reference itself cannot run this case: its sub-engines share one model and therefore one
    LSTT memory (layers/transformer.py:1000-1007 lives on the shared module), so the second
    engine re-fuses already fused memories and raises (tests/golden/make_golden.py note).  This
    class restates what the wrapper computes when every sub-engine keeps its own state, which is
    also what rmem_amd.engine.DeAOTInferEngine does."""

    def __init__(self, model, long_term_mem_gap: int = 5):
        self.model, self.cfg = model, model.cfg
        self.long_term_mem_gap = long_term_mem_gap
        self.max_obj = self.cfg.MODEL_MAX_OBJ_NUM
        self.engines: List[OracleDeAOTEngine] = []

    def restart_engine(self):                                   # aot_engine.py:598-602
        self.engines = []

    def separate_mask(self, mask):                              # aot_engine.py:604-618 (label masks)
        if len(self.engines) == 1:
            return [mask]
        out = []
        for idx in range(len(self.engines)):
            start_id, end_id = idx * self.max_obj + 1, (idx + 1) * self.max_obj
            fg = ((mask >= start_id) & (mask <= end_id)).float()
            out.append((fg * mask - start_id + 1) * fg)
        return out

    def soft_logit_aggregation(self, all_logits):               # aot_engine.py:650-673
        if len(all_logits) == 1:
            return all_logits[0]
        probs = [torch.softmax(lg, dim=1) for lg in all_logits]
        bg = torch.prod(torch.cat([p[:, 0:1] for p in probs], dim=1), dim=1, keepdim=True)
        merged = torch.cat([bg] + [p[:, 1:1 + self.max_obj] for p in probs], dim=1).clamp(1e-5, 1 - 1e-5)
        return torch.logit(merged)

    def add_reference_frame(self, img, mask, obj_nums, frame_step=-1):      # :675-702
        n = obj_nums[0] if isinstance(obj_nums, list) else obj_nums
        need = max(-(-int(n) // self.max_obj), 1)
        while len(self.engines) < need:
            self.engines.append(OracleDeAOTEngine(self.model, self.long_term_mem_gap))
        for e, m in zip(self.engines, self.separate_mask(mask)):
            e.add_reference_frame(img, m, obj_nums=[self.max_obj], frame_step=frame_step)
        self.input_size_2d = self.engines[0].input_size_2d

    def match_propogate_one_frame(self, img, mask=None, output_size=None):  # :704-712
        return self.soft_logit_aggregation([e.match_propogate_one_frame(img, output_size=output_size)
                                            for e in self.engines])

    def update_memory(self, curr_mask):                         # :714-720
        for e, m in zip(self.engines, self.separate_mask(curr_mask)):
            e.update_memory(m)


def run_clip(engine, imgs, label0, out_hw=None):
    """The evaluator's per-frame protocol (networks/managers/evaluator.py:384-441,
    518-523): reference frame, then per frame match -> softmax -> argmax -> nearest
    resize -> update_memory.  Returns list of int64 label maps [H0,W0] (frame 1..)."""
    H, W = imgs[0].shape[2:]
    if out_hw is None:
        out_hw = (H, W)
    engine.restart_engine()
    engine.add_reference_frame(imgs[0], label0, obj_nums=[int(label0.max().item())], frame_step=0)
    labels = []
    for t in range(1, len(imgs)):
        logit = engine.match_propogate_one_frame(imgs[t], output_size=out_hw)
        prob = torch.softmax(logit, dim=1)
        pred = torch.argmax(prob, dim=1, keepdim=True).float()
        labels.append(pred[0, 0].long().cpu())
        cur = F.interpolate(pred, size=engine.input_size_2d, mode="nearest")
        engine.update_memory(cur)
    return labels
