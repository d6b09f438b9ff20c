"""Name-keyed synthetic weights and synthetic clips.

Every parameter is drawn from ``numpy.random.RandomState(crc32(name))`` with a
documented scale, so this container (oracle / fixture generation against the
imported reference) and the GPU box build bit-identical tensors without shipping
weights.  Scales are chosen so that the encoder
embedding that feeds the LSTT is O(1) and the memory path demonstrably moves the
mask (default init gives std~20 features that drown the LSTT, SURVEY.md section 7
"Hard parts").
"""
from __future__ import annotations

import zlib

import numpy as np
import torch


def _rs(name: str, salt: int = 0) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(name.encode()) + salt) & 0x7FFFFFFF)


def synth_tensor(name: str, shape, salt: int = 0) -> torch.Tensor:
    """Deterministic value for parameter/buffer ``name`` of ``shape``."""
    shape = tuple(shape)
    rs = _rs(name, salt)
    leaf = name.split(".")[-1]

    def normal(std):
        return torch.from_numpy(rs.standard_normal(shape).astype(np.float32) * np.float32(std))

    if name.startswith("encoder."):
        if leaf == "running_mean":
            return normal(0.05)
        if leaf == "running_var":
            return torch.from_numpy((1.0 + 0.1 * rs.rand(*shape)).astype(np.float32))
        if leaf == "bias":
            return normal(0.05)
        if leaf == "weight" and len(shape) == 1:      # FrozenBN scale
            base = 0.35 if ".bn3." in name else 1.0   # damp residual growth
            return torch.from_numpy((base * (1.0 + 0.1 * (rs.rand(*shape) - 0.5))).astype(np.float32))
        if "relative_position_bias_table" in name:     # Swin window attention bias
            return normal(0.5)
        if len(shape) == 2:                             # Swin linears (qkv, proj, mlp, reduction)
            return normal(1.0 / shape[1] ** 0.5)
        fan_in = shape[1] * shape[2] * shape[3]
        return normal((2.0 / fan_in) ** 0.5)
    if name in ("cur_pos_emb", "mem_pos_emb"):
        return normal(0.5)
    if name.startswith("patch_wise_id_bank"):
        if leaf == "bias":
            return normal(0.02)
        return normal(0.05)
    if len(shape) == 1:
        if leaf == "weight":                           # LN / GN gamma
            return torch.from_numpy((1.0 + 0.2 * (rs.rand(*shape) - 0.5)).astype(np.float32))
        if name == "decoder.conv_4x.gn.bias":
            return normal(0.05) - 1.0                  # sparse last activations -> varied label map
        return normal(0.05)                            # biases / betas
    if "relative_emb_k" in name:
        return normal(0.15)
    if "dw_conv" in name or ("activation.conv" in name):
        return normal(0.25)
    fan_in = int(np.prod(shape[1:]))
    gain = 1.0
    if "encoder_projector" in name:
        gain = 0.2
    if name.startswith("decoder."):
        # keep the encoder shortcuts from drowning the LSTT embedding in the FPN
        gain = 0.15 if ".adapter_" in name or "decoder.adapter_" in name else 1.4
    t = normal(gain / fan_in ** 0.5)
    # sharpen attention (trained models are peaky; near-uniform attention would hide
    # layout/indexing bugs): scale the query/key projections.
    if name.endswith("linear_QV.weight"):
        t[:128] *= 2.0
    if name.endswith("linear_QK.weight"):
        t *= 2.0
    if name.endswith(".linear_Q.weight") or name.endswith(".linear_K.weight"):   # AOT blocks (8 x 32 heads)
        t *= 2.0
    return t


@torch.no_grad()
def load_synthetic_weights(model: torch.nn.Module, salt: int = 0) -> None:
    """Overwrite every entry of ``model.state_dict()`` in place, keyed by name."""
    sd = model.state_dict()
    for k, v in sd.items():
        if v.dtype.is_floating_point:
            v.copy_(synth_tensor(k, v.shape, salt).to(v.dtype))


def synth_clip(seed: int, frames: int, height: int, width: int, n_obj: int = 3,
               device="cpu"):
    """Synthetic clip: img_t = base + 0.1*t*noise.

    Returns (imgs [F,1,3,H,W] float32 list, label0 [1,1,H,W] float32 with ids 1..n_obj).
    """
    g = torch.Generator().manual_seed(int(seed))
    base = torch.randn(1, 3, height, width, generator=g)
    noise = torch.randn(1, 3, height, width, generator=g)
    imgs = [(base + 0.1 * t * noise).to(device) for t in range(frames)]
    label = torch.zeros(1, 1, height, width)
    for o in range(n_obj):
        y0 = int(height * (0.1 + 0.25 * o)); y1 = int(height * (0.3 + 0.25 * o))
        x0 = int(width * (0.15 + 0.2 * o)); x1 = int(width * (0.45 + 0.2 * o))
        label[:, :, y0:y1, x0:x1] = o + 1
    return imgs, label.to(device)
