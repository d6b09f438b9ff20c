"""Checkpoint compatibility: the remap rules of the reference's ``load_network``
(/root/reference/aot_plus/utils/checkpoint.py:75-101) so that published RMem / AOT
checkpoints load into ``rmem_amd.model`` (whose ``state_dict()`` keys are the reference's).

Rules (same order as the reference):
  1. payload under 'state_dict', else 'model', else the dict itself (:78-83);
  2. a >=3-D tensor whose dim-0 matches and whose dim-1 is exactly one short of the model's
     is written into ``[:, :-1]`` of the model tensor, the last input channel keeps the
     model's value (:88-90) -- this is how 11-channel ``patch_wise_id_bank`` weights load into
     the 12-channel (ignore-token) bank of stage pre_vost;
  3. exact name + shape match -> loaded (:91-92);
  4. keys with a 'module.' prefix (DataParallel) are retried without it (:93-95);
  5. everything else is returned in the ``removed`` list (the reference silently drops
     shape mismatches; here they are reported too).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch


def remap_state_dict(model_sd: Dict[str, torch.Tensor], ckpt) -> Tuple[Dict[str, torch.Tensor], List[str]]:
    if isinstance(ckpt, dict) and "state_dict" in ckpt:
        src = ckpt["state_dict"]
    elif isinstance(ckpt, dict) and "model" in ckpt:
        src = ckpt["model"]
    else:
        src = ckpt
    out = {k: v.clone() for k, v in model_sd.items()}
    removed: List[str] = []
    for k, v in src.items():
        if k in out and v.dim() > 2 and v.shape[0] == out[k].shape[0] and v.shape[1] == out[k].shape[1] - 1:
            out[k][:, :-1] = v.to(out[k].dtype)
            continue
        if k in out and v.shape == out[k].shape:
            out[k] = v.to(out[k].dtype)
        elif k.startswith("module.") and k[7:] in out and v.shape == out[k[7:]].shape:
            out[k[7:]] = v.to(out[k[7:]].dtype)
        else:
            removed.append(k)
    return out, removed


def load_network(net: torch.nn.Module, path_or_ckpt, device=None):
    """Returns (net, removed_keys).  ``path_or_ckpt``: file path or an already loaded object."""
    ckpt = torch.load(path_or_ckpt, map_location="cpu") if isinstance(path_or_ckpt, (str, bytes)) else path_or_ckpt
    sd, removed = remap_state_dict(net.state_dict(), ckpt)
    net.load_state_dict(sd)
    if device is not None:
        net = net.to(device)
    # engines pack the LSTT / ID-bank weights into planes when they are built: tell them to re-pack
    object.__setattr__(net, "_weights_version", net.__dict__.get("_weights_version", 0) + 1)
    if net.__dict__.get("_enc_infer") is not None:      # folded inference encoder is stale now (folds again: new version)
        net.optimize_for_inference(True)
    return net, removed
