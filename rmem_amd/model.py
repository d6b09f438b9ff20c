"""VOS model container: PyTorch encoder/decoder + parameter holders for the LSTT.

``build_vos_model(name, cfg)`` mirrors the reference factory
(/root/reference/aot_plus/networks/models/__init__.py:5-12).  The returned module has
reference-identical ``state_dict()`` keys so reference
checkpoints load by name, but the LSTT sub-modules are *parameter holders only*:
their forward pass is executed by the HIP kernels in ``rmem_amd.csrc`` through
``rmem_amd.lstt.DeAOTLSTT`` (the engine calls that, never ``nn.Module.forward``).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .nets import ResNet50Encoder, FPNHead, SwinBEncoder


class _DW(nn.Module):
    """Holder for DWConv2d.conv (layers/basic.py:38-47): 5x5 depth-wise, no bias."""

    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 5, padding=2, groups=c, bias=False)


class _GP(nn.Module):
    """Holder for GatedPropagation params (layers/attention.py:93-137)."""

    def __init__(self, d_qk, d_vu, d_att, use_linear):
        super().__init__()
        e = d_vu * 2
        if use_linear:
            self.linear_QK = nn.Linear(d_qk, d_att)
            self.linear_V1 = nn.Linear(d_vu // 2, e // 2)
            self.linear_V2 = nn.Linear(d_vu // 2, e // 2)
            self.linear_U1 = nn.Linear(d_vu // 2, e // 2)
            self.linear_U2 = nn.Linear(d_vu // 2, e // 2)
        self.dw_conv = _DW(e)
        self.projection = nn.Linear(e, d_vu)


class _LocalGP(nn.Module):
    """Holder for LocalGatedPropagation params (layers/attention.py:220-287)."""

    def __init__(self, d_vu, d_att, max_dis=7):
        super().__init__()
        e = d_vu * 2
        win = 2 * max_dis + 1
        self.relative_emb_k = nn.Conv2d(d_att, win * win, 1)
        self.dw_conv = _DW(e)
        self.projection = nn.Linear(e, d_vu)


class _GPMLayer(nn.Module):
    """Holder for GatedPropagationModule params (layers/transformer.py:1010-1082)."""

    def __init__(self, d_model, layer_idx):
        super().__init__()
        d_att = d_model // 2
        e = d_model * 2
        self.norm1 = nn.LayerNorm(d_model)
        self.linear_QV = nn.Linear(d_model, d_att + e)
        self.linear_U = nn.Linear(d_model, e)
        if layer_idx == 0:
            self.linear_ID_V = nn.Linear(d_model, e)
        else:
            self.id_norm1 = nn.LayerNorm(d_model)
            self.linear_ID_V = nn.Linear(d_model * 2, e)
            self.linear_ID_U = nn.Linear(d_model, e)
        self.long_term_attn = _GP(d_model, e, d_att, use_linear=False)
        self.short_term_attn = _LocalGP(e, d_att)
        self.norm2 = nn.LayerNorm(d_model)
        self.id_norm2 = nn.LayerNorm(d_model)
        self.self_attn = _GP(e, e, d_att, use_linear=True)


class _GN1D(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.gn = nn.GroupNorm(groups, c)


class _DualBranchGPM(nn.Module):
    """Holder for DualBranchGPM params (layers/transformer.py:700-763)."""

    def __init__(self, num_layers, d_model):
        super().__init__()
        self.layers = nn.ModuleList(_GPMLayer(d_model, i) for i in range(num_layers))
        self.decoder_norms = nn.ModuleList([_GN1D(d_model * 2, 2)])


class DeAOT(nn.Module):
    """DeAOT + RMem model (models/deaot.py:10-69, models/aot.py:12-103)."""

    def __init__(self, cfg):
        super().__init__()
        if cfg.MODEL_ENCODER != "resnet50":
            raise NotImplementedError(cfg.MODEL_ENCODER)
        if cfg.MODEL_ATT_HEADS != 1 or cfg.MODEL_SELF_HEADS != 1:
            raise NotImplementedError("DeAOT hot path is built for 1 head (default_deaot.py:14-15)")
        self.cfg = cfg
        self.max_obj_num = cfg.MODEL_MAX_OBJ_NUM
        d = cfg.MODEL_ENCODER_EMBEDDING_DIM
        self.encoder = ResNet50Encoder()
        self.encoder_projector = nn.Conv2d(cfg.MODEL_ENCODER_DIM[-1], d, 1)
        self.LSTT = _DualBranchGPM(cfg.MODEL_LSTT_NUM, d)
        self.decoder = FPNHead(d * 2, cfg.MODEL_MAX_OBJ_NUM + 1, hidden_dim=d,
                               shortcut_dims=cfg.MODEL_ENCODER_DIM,
                               align_corners=cfg.MODEL_ALIGN_CORNERS,
                               decode_intermediate_input=False)
        id_dim = cfg.MODEL_MAX_OBJ_NUM + (2 if cfg.MODEL_IGNORE_TOKEN else 1)
        if cfg.MODEL_ALIGN_CORNERS:
            self.patch_wise_id_bank = nn.Conv2d(id_dim, d, 17, stride=16, padding=8)
        else:
            self.patch_wise_id_bank = nn.Conv2d(id_dim, d, 16, stride=16, padding=0)
        self.id_norm = nn.LayerNorm(d)
        self.use_temporal_pe = cfg.USE_TEMPORAL_POSITIONAL_EMBEDDING
        if not (self.use_temporal_pe and cfg.TEMPORAL_POSITIONAL_EMBEDDING_SLOT_4):
            raise NotImplementedError("hot path is built for the 4-slot temporal PE (r50_deaotl.py:14-16)")
        self.cur_pos_emb = nn.Parameter(torch.zeros(1, d // 2))
        self.mem_pos_emb = nn.Parameter(torch.zeros(4, d // 2))

    def optimize_for_inference(self, fold_bn: bool = True, force: bool = False):
        """Build the inference-time encoder (FrozenBN folded into the convs).  Call after the weights are loaded.

        Idempotent per (fold_bn, weights version, device): every engine calls it when it is built, and an engine
        built EARLIER holds hipGraphs that replay the folded tensors -- folding again would free the memory those
        graphs read (a second driver on the same model used to corrupt the first one's encoder that way).  It folds
        again when the weights changed: rmem_amd.checkpoint.load_network() bumps `_weights_version`; ENCODER weights
        written in place by hand (load_state_dict, copy_) are noticed through the tensors' version counters -- but only
        here, i.e. when the next engine is built; force=True folds in any case.  The last two bump the version themselves so
        that existing engines re-pack their LSTT planes and re-capture their graphs at their next clip
        (engine.py / batched.py: _stale_weights).  NOT covered: in-place edits of LSTT / decoder / ID-bank weights without
        load_network(), and edits made while engines exist and no new one is built -- call
        optimize_for_inference(force=True) (or load_network) after changing weights by hand."""
        wv = self.__dict__.get("_weights_version", 0)
        dev = str(next(self.parameters()).device)
        have = self.__dict__.get("_enc_infer_state")
        # weights written in place since the last fold (load_state_dict / copy_ without load_network): every tensor carries
        # a version counter that in-place writes bump -- a changed (storage, version) list means the folded copy is stale
        import itertools
        mark = hash(tuple((t.data_ptr(), t._version) for t in itertools.chain(self.encoder.parameters(), self.encoder.buffers())))
        changed = have is not None and self.__dict__.get("_enc_infer_mark") != mark
        if not force and not changed and have == (bool(fold_bn), wv, dev):
            return self
        if (force or changed) and have is not None and have[1] == wv:        # (load_network has bumped the version already)
            wv += 1
            object.__setattr__(self, "_weights_version", wv)
        object.__setattr__(self, "_enc_infer_mark", mark)
        # kept out of nn.Module registration so that state_dict() keeps the reference's keys
        object.__setattr__(self, "_enc_infer", self.encoder.folded() if fold_bn else None)
        object.__setattr__(self, "_enc_infer_state", (bool(fold_bn), wv, dev))
        return self

    # -- pass-throughs to PyTorch (models/aot.py:116-134, deaot.py:57-63)
    def encode_image(self, img):
        enc = self.__dict__.get("_enc_infer") or self.encoder
        xs = enc(img)
        xs[-1] = self.encoder_projector(xs[-1])
        if xs[0].is_cuda and not torch.is_grad_enabled() and hasattr(self.decoder, "adapter_convs"):
            # the decoder's skip adapters read encoder features only: run them with the encoder pass
            xs = FeatureList(xs)
            xs.adapters = self.decoder.adapter_convs(xs)
        return xs

    def decode_id_logits(self, lstt_emb_nc, shortcuts):
        """lstt_emb_nc: [N, 2d] token-major tensor (the LSTT output)."""
        n, _, h, w = shortcuts[-1].shape
        emb = lstt_emb_nc.view(h, w, n, -1).permute(2, 3, 0, 1)
        return self.decoder([shortcuts[-1], emb], shortcuts)


class FeatureList(list):
    """Encoder pyramid [4x, 8x, 16x, projected 16x] plus, on the GPU inference path, the decoder's
    bias-free skip-adapter convolutions of it (`adapters`, see nets/fpn.py:FPNHead.adapter_convs)."""
    adapters = None


class _MHA(nn.Module):
    """Holder for MultiheadAttention params (layers/attention.py:8-26)."""

    def __init__(self, d, use_linear):
        super().__init__()
        if use_linear:
            self.linear_Q = nn.Linear(d, d)
            self.linear_K = nn.Linear(d, d)
            self.linear_V = nn.Linear(d, d)
        self.projection = nn.Linear(d, d)


class _GNAct(nn.Module):
    """Holder for GNActDWConv2d params (layers/basic.py:15-25)."""

    def __init__(self, c):
        super().__init__()
        self.gn = nn.GroupNorm(32, c)
        self.conv = nn.Conv2d(c, c, 5, padding=2, groups=c, bias=False)


class _AOTBlock(nn.Module):
    """Holder for SimplifiedTransformerBlock params, linear_q=False
    (layers/transformer.py:466-517)."""

    def __init__(self, d, ff=1024):
        super().__init__()
        self.norm1 = nn.LayerNorm(d)
        self.self_attn = _MHA(d, True)
        self.norm2 = nn.LayerNorm(d)
        self.linear_Q = nn.Linear(d, d)
        self.linear_V = nn.Linear(d, d)
        self.linear_QMem = nn.Linear(d, d)
        self.linear_VMem = nn.Linear(d, d)
        self.norm4 = nn.LayerNorm(d)
        self.linear_KMem = nn.Linear(d, d)     # present in checkpoints, never used (:494)
        self.long_term_attn = _MHA(d, False)
        self.short_term_attn = _MHA(d, False)
        self.norm3 = nn.LayerNorm(d)
        self.linear1 = nn.Linear(d, ff)
        self.activation = _GNAct(ff)
        self.linear2 = nn.Linear(ff, d)


class _LSTT(nn.Module):
    """Holder for LongShortTermTransformer params (layers/transformer.py:133-197)."""

    def __init__(self, num_layers, d):
        super().__init__()
        self.layers = nn.ModuleList(_AOTBlock(d) for _ in range(num_layers))
        self.decoder_norms = nn.ModuleList(nn.LayerNorm(d) for _ in range(num_layers))


class AOT(nn.Module):
    """AOT + RMem model (models/aot.py:12-103), 8 heads x 32, stage pre_vost
    (MODEL_LINEAR_Q=False -> norm4 short-term variant, 12-channel ID bank)."""

    def __init__(self, cfg):
        super().__init__()
        if cfg.MODEL_ENCODER not in ("resnet50", "swin_base"):
            raise NotImplementedError(cfg.MODEL_ENCODER)
        if cfg.MODEL_ATT_HEADS != 8 or cfg.MODEL_SELF_HEADS != 8 or cfg.MODEL_LINEAR_Q:
            raise NotImplementedError("AOT hot path is built for 8 heads, MODEL_LINEAR_Q=False")
        self.cfg = cfg
        self.max_obj_num = cfg.MODEL_MAX_OBJ_NUM
        d = cfg.MODEL_ENCODER_EMBEDDING_DIM
        self.encoder = ResNet50Encoder() if cfg.MODEL_ENCODER == "resnet50" else SwinBEncoder()
        self.encoder_projector = nn.Conv2d(cfg.MODEL_ENCODER_DIM[-1], d, 1)
        self.LSTT = _LSTT(cfg.MODEL_LSTT_NUM, d)
        self.decoder = FPNHead(d * (cfg.MODEL_LSTT_NUM + 1), cfg.MODEL_MAX_OBJ_NUM + 1, hidden_dim=d,
                               shortcut_dims=cfg.MODEL_ENCODER_DIM, align_corners=cfg.MODEL_ALIGN_CORNERS,
                               decode_intermediate_input=True)
        id_dim = cfg.MODEL_MAX_OBJ_NUM + (2 if cfg.MODEL_IGNORE_TOKEN else 1)
        if cfg.MODEL_ALIGN_CORNERS:
            self.patch_wise_id_bank = nn.Conv2d(id_dim, d, 17, stride=16, padding=8)
        else:
            self.patch_wise_id_bank = nn.Conv2d(id_dim, d, 16, stride=16, padding=0)
        self.use_temporal_pe = True
        self.cur_pos_emb = nn.Parameter(torch.zeros(1, d))
        self.mem_pos_emb = nn.Parameter(torch.zeros(4, d))

    optimize_for_inference = DeAOT.optimize_for_inference
    encode_image = DeAOT.encode_image

    def decode_id_logits(self, lstt_embs, shortcuts):
        """lstt_embs: list of 3 token-major [N, d] tensors (models/aot.py:136-142)."""
        n, _, h, w = shortcuts[-1].shape
        ins = [shortcuts[-1]] + [e.view(h, w, n, -1).permute(2, 3, 0, 1) for e in lstt_embs]
        return self.decoder(ins, shortcuts)


def build_vos_model(name, cfg, **kwargs):
    if name == "deaot":
        return DeAOT(cfg, **kwargs)
    if name == "aot":
        return AOT(cfg, **kwargs)
    raise NotImplementedError(name)
