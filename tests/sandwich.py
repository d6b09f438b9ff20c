"""Test helper: engines whose ONLY GPU arithmetic is the hot path (rmem_amd/csrc).

``SandwichInferEngine`` is rmem_amd.engine.DeAOTInferEngine with the encoder and the FPN decoder run by
the CPU copy of the same model (PyTorch-CPU fp32: deterministic, no MIOpen) around the HIP LSTT, ID
assignment and memory update.  With it a label map is a function of the inputs and of rmem_amd/csrc
alone, so statements that MIOpen's process-level rounding makes "almost always true" on the product
engines (rank-count invariance, closed-loop equality with the oracle) can be asserted EXACTLY for the
code this repository owns.  Same trick as test_480p_lstt_isolated_from_miopen."""
import copy

import torch
import torch.nn.functional as F

from rmem_amd.engine import DeAOTEngine, DeAOTInferEngine


class SandwichEngine(DeAOTEngine):
    def __init__(self, cpu_model, gpu_model, gpu_id=0, long_term_mem_gap=9999, nsplit=3):
        super().__init__(gpu_model, gpu_id, long_term_mem_gap, 1, nsplit, use_graphs=False)
        object.__setattr__(self, "_cpu_model", cpu_model)
        self._enc_cpu = None

    def _encode_cpu(self, img):
        with torch.no_grad():
            self._enc_cpu = self._cpu_model.encode_image(img.detach().float().cpu())
        dev = next(self.AOT.parameters()).device
        return [x.to(dev) for x in self._enc_cpu]

    @torch.no_grad()
    def add_reference_frame(self, img=None, mask=None, frame_step=-1, obj_nums=None, img_embs=None):
        return super().add_reference_frame(img, mask, frame_step, obj_nums, img_embs=self._encode_cpu(img))

    @torch.no_grad()
    def match_propogate_one_frame(self, img=None, img_embs=None, mask=None, output_size=None, next_img=None):
        return super().match_propogate_one_frame(img, img_embs=self._encode_cpu(img), mask=mask, output_size=output_size)

    def decode_current_logits(self, enc, lstt_out, output_size=None):
        logits = self._cpu_model.decode_id_logits(lstt_out.detach().cpu(), self._enc_cpu)   # CPU decoder on the HIP LSTT output
        for batch_idx, obj_num in enumerate(self.obj_nums):
            logits[batch_idx, (obj_num + 1):] = -1e10
        dev = lstt_out.device
        self.pred_id_logits = logits.to(dev)
        if output_size is not None:
            logits = F.interpolate(logits, size=output_size, mode="bilinear", align_corners=self.align_corners)
        return logits.to(dev)


class SandwichInferEngine(DeAOTInferEngine):
    supports_prefetch = False

    def __init__(self, cpu_model, device="cuda:0", gpu_id=0, long_term_mem_gap=9999, nsplit=3, gpu_model=None):
        gm = gpu_model if gpu_model is not None else copy.deepcopy(cpu_model).to(device)
        object.__setattr__(self, "_cpu_model_ref", cpu_model)
        super().__init__(gm, gpu_id, long_term_mem_gap, nsplit=nsplit, fold_bn=False, use_graphs=False)

    def _new_engine(self):
        return SandwichEngine(self._cpu_model_ref, self.AOT, self.gpu_id, self.long_term_mem_gap, self.nsplit)
