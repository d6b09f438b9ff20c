"""Seeded input generators shared by make_golden.py (reference side) and the tests
(oracle / HIP side).  numpy RandomState streams are stable across platforms, so the
fixtures only need to hold the reference's *outputs*."""
from __future__ import annotations

import numpy as np
import torch

BLOCK_CASES = [  # (layer, T, h, w, ref_frame)
    (0, 1, 5, 7, False), (1, 2, 5, 7, False), (1, 4, 8, 11, False),
    (2, 5, 8, 11, False), (1, 8, 5, 7, False), (0, 1, 8, 11, True), (2, 1, 5, 7, True),
    (1, 3, 9, 13, False),
]


def block_case_name(layer, T, h, w, ref_frame):
    return f"block_l{layer}_T{T}_{h}x{w}{'_ref' if ref_frame else ''}"


def block_inputs(layer, T, h, w, ref_frame):
    rs = np.random.RandomState(1000 + 17 * layer + T * 3 + h + (500 if ref_frame else 0))
    n = h * w
    r = lambda *s: torch.from_numpy(rs.standard_normal(s).astype(np.float32))
    d = dict(tgt=r(n, 256), tgt_id=None if layer == 0 else r(n, 256) * 0.7)
    # key-like tensors get the scale the synthetic model produces (std ~1.9)
    d.update(bank_K=r(T, n, 128) * 1.5, bank_V=r(T, n, 512) * 0.5, bank_IDV=r(T, n, 512) * 0.5,
             short_K=r(n, 128) * 1.5, short_V=r(n, 512) * 0.5, short_IDV=r(n, 512) * 0.5,
             id_emb=r(n, 256))
    return d


IDASSIGN_CASES = [(97, 129), (65, 81), (49, 33)]


def idassign_label(H, W):
    rs = np.random.RandomState(H * 7 + W)
    coarse = torch.from_numpy(rs.randint(0, 11, (1, 1, H // 8 + 1, W // 8 + 1)).astype(np.float32))
    label = torch.nn.functional.interpolate(coarse, size=(H, W), mode="nearest")
    label[:, :, 10:20, 5:25] = 255
    return label


AOT_BLOCK_CASES = [  # (layer, T, h, w, ref_frame)
    (0, 1, 5, 7, False), (1, 2, 6, 7, False), (2, 4, 8, 11, False), (1, 5, 6, 9, False),
    (0, 1, 8, 11, True), (2, 1, 5, 7, True),
]


def aot_block_case_name(layer, T, h, w, ref_frame):
    return f"aot_block_l{layer}_T{T}_{h}x{w}{'_ref' if ref_frame else ''}"


def aot_block_inputs(layer, T, h, w, ref_frame):
    rs = np.random.RandomState(3000 + 13 * layer + T * 5 + h + (500 if ref_frame else 0))
    n = h * w
    r = lambda *s: torch.from_numpy(rs.standard_normal(s).astype(np.float32))
    return dict(tgt=r(n, 256), bank_K=r(T, n, 256) * 1.5, bank_V=r(T, n, 256) * 0.7,
                short_K=r(n, 256) * 1.5, short_V=r(n, 256) * 0.7, id_emb=r(n, 256) * 0.5)


def multiobj_label(H, W, n_obj=12):
    """n_obj rectangles on a 4-column grid (ids 1..n_obj) -> [1,1,H,W] float."""
    lab = torch.zeros(1, 1, H, W)
    rows = (n_obj + 3) // 4
    for o in range(n_obj):
        r, c = o // 4, o % 4
        y0, y1 = int(H * (0.05 + 0.9 * r / rows)), int(H * (0.05 + 0.9 * (r + 0.8) / rows))
        x0, x1 = int(W * (0.03 + 0.24 * c)), int(W * (0.03 + 0.24 * c + 0.19))
        lab[:, :, y0:y1, x0:x1] = o + 1
    return lab


# ---- load_network cases (utils/checkpoint.py:75-101): a small module with the key shapes the rules look at (a 4-D
# "id bank" whose input channels may be one short, 2-D / 1-D tensors, nested names) and the payload variants
def ckpt_toy_net(seed=0):
    g = torch.Generator().manual_seed(seed)
    net = torch.nn.Sequential()
    net.add_module("patch_wise_id_bank", torch.nn.Conv2d(12, 8, 3, bias=True))
    net.add_module("proj", torch.nn.Linear(6, 5))
    net.add_module("norm", torch.nn.LayerNorm(5))
    blk = torch.nn.Sequential()
    blk.add_module("conv", torch.nn.Conv2d(4, 4, 1, bias=False))
    blk.add_module("bn", torch.nn.BatchNorm2d(4))
    net.add_module("block", blk)
    with torch.no_grad():
        for p_ in net.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g))
        for b_ in net.buffers():
            if b_.dtype.is_floating_point:
                b_.copy_(torch.rand(b_.shape, generator=g) + 0.5)
    return net


def ckpt_cases():
    """name -> checkpoint object as torch.save would hold it."""
    src = ckpt_toy_net(1).state_dict()
    r = lambda *s, seed=5: torch.randn(*s, generator=torch.Generator().manual_seed(seed))
    pre = lambda d: {"module." + k: v.clone() for k, v in d.items()}
    bank11 = {k: v.clone() for k, v in src.items()}
    bank11["patch_wise_id_bank.weight"] = src["patch_wise_id_bank.weight"][:, :11].clone()
    mism = {k: v.clone() for k, v in src.items()}
    mism["proj.weight"] = r(5, 7)                         # 2-D, wrong inner size
    mism["block.conv.weight"] = r(4, 5, 1, 1)             # 4-D, dim-1 one LONGER (rule 2 wants one shorter)
    mism["block.bn.running_mean"] = r(3)
    mism["not_in_model.weight"] = r(2, 2)
    pmism = pre(mism)
    pmism["module.patch_wise_id_bank.weight"] = src["patch_wise_id_bank.weight"][:, :11].clone()
    mixed = {k: v.clone() for k, v in src.items() if k.startswith("proj")}
    mixed.update(pre({k: v for k, v in src.items() if k.startswith("block")}))
    mixed["module.module.norm.weight"] = src["norm.weight"].clone()      # doubly prefixed: stripped once, then unknown
    return {
        "plain": {k: v.clone() for k, v in src.items()},
        "state_dict_key": {"state_dict": {k: v.clone() for k, v in src.items()}, "optimizer": {"state": {}}},
        "model_key": {"model": {k: v.clone() for k, v in src.items()}},
        "both_keys": {"state_dict": {k: v.clone() for k, v in src.items()}, "model": {"proj.weight": r(5, 6)}},
        "module_prefixed": pre(src),
        "bank11": bank11,
        "bank11_prefixed": pre(bank11),
        "mismatch": mism,
        "mismatch_prefixed": pmism,
        "mixed": mixed,
    }
