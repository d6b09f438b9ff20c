"""Checkpoint compatibility: the remap rules of the reference's ``load_network``
(/root/reference/aot_plus/utils/checkpoint.py:75-101) and the evaluator's choice of the checkpoint file
(networks/managers/evaluator.py:59-110), so that published RMem / AOT checkpoints load into
``rmem_amd.model`` (whose ``state_dict()`` keys are the reference's).

``remap_state_dict`` -- rules in the reference's order, pinned by tests/golden/load_network_cases.* (the reference's own
function run over ten payload variants, make_golden.py:gen_load_network_cases):
  1. payload under 'state_dict', else 'model', else the dict itself (:78-83);
  2. a >=3-D tensor under a key OF THE MODEL whose dim-0 matches and whose dim-1 is exactly one short of the model's is
     written into ``[:, :-1]`` of the model tensor, the last input channel keeps the model's value (:88-90) -- this is
     how 11-channel ``patch_wise_id_bank`` weights load into the 12-channel (ignore-token) bank of stage pre_vost;
  3. exact name + shape match -> loaded (:91-92);
  4. a key with a 'module.' prefix (DataParallel) is retried ONCE without it: loaded on a name + shape match, and
     otherwise dropped WITHOUT a report (:93-95) -- rule 2 is not retried for it, so a prefixed 11-channel bank is not
     loaded at all;
  5. every other key (not in the model, or in the model with another shape) is returned in ``removed`` (:96-97).
``dropped`` (optional list) additionally collects the keys rule 4 drops silently -- the reference has no counterpart.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch


def remap_state_dict(model_sd: Dict[str, torch.Tensor], ckpt,
                     dropped: Optional[List[str]] = None) -> Tuple[Dict[str, torch.Tensor], List[str]]:
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
        elif k[:7] == "module.":
            if k[7:] in out and v.shape == out[k[7:]].shape:
                out[k[7:]] = v.to(out[k[7:]].dtype)
            elif dropped is not None:
                dropped.append(k)
        else:
            removed.append(k)
    return out, removed


def load_network(net: torch.nn.Module, path_or_ckpt, device=None, dropped: Optional[List[str]] = None):
    """Returns (net, removed_keys).  ``path_or_ckpt``: file path or an already loaded object."""
    ckpt = torch.load(path_or_ckpt, map_location="cpu") if isinstance(path_or_ckpt, (str, bytes, os.PathLike)) else path_or_ckpt
    sd, removed = remap_state_dict(net.state_dict(), ckpt, dropped)
    net.load_state_dict(sd)
    if device is not None:
        net = net.to(device)
    # engines pack the LSTT / ID-bank weights into planes when they are built: tell them to re-pack
    object.__setattr__(net, "_weights_version", net.__dict__.get("_weights_version", 0) + 1)
    if net.__dict__.get("_enc_infer") is not None:      # folded inference encoder is stale now (folds again: new version)
        net.optimize_for_inference(True)
    return net, removed


def select_checkpoint(cfg) -> Tuple[str, Optional[str]]:
    """Which checkpoint file an evaluation loads: ``Evaluator.process_pretrained_model``
    (networks/managers/evaluator.py:59-110).  Returns (ckpt label, path); path None = nothing is loaded.

      * ``cfg.TEST_CKPT_PATH == 'test'``: label 'test', nothing loaded (:62-65);
      * ``cfg.TEST_CKPT_PATH is None``: the step is ``cfg.TEST_CKPT_STEP`` if set, else the LARGEST step among the files
        of ``cfg.DIR_CKPT`` (step = the integer between the last '_' and the first '.' of the file name, :71-78; an
        empty directory is an error, :80-82).  With ``cfg.TEST_EMA`` the FILE is then taken from
        ``<cfg.DIR_RESULT>/ema_ckpt`` (:84-85) -- the step was still chosen from the listing of the plain directory,
        as in the reference.  Path = ``<dir>/save_step_<step>.pth`` (:86-89);
      * otherwise the given path, label 'unknown' (:99-101).
    Like the reference, the choice is also written back to ``cfg.DIR_CKPT`` / ``cfg.TEST_CKPT_PATH`` -- so the function is
    not idempotent: a second call finds TEST_CKPT_PATH set and returns ('unknown', that path) without listing anything."""
    if cfg.TEST_CKPT_PATH == "test":
        return "test", None
    if cfg.TEST_CKPT_PATH is None:
        if getattr(cfg, "TEST_CKPT_STEP", None) is not None:
            ckpt = str(cfg.TEST_CKPT_STEP)
        else:
            names = os.listdir(cfg.DIR_CKPT)
            if not names:
                raise FileNotFoundError(f"No checkpoint in {cfg.DIR_CKPT}.")
            # the reference parses EVERY name (a stray .DS_Store or temporary file is a bare ValueError there); same rule
            # -- the integer between the last '_' and the first '.' -- applied to the names it can apply to
            steps = []
            for x in names:
                try:
                    steps.append(int(x.split("_")[-1].split(".")[0]))
                except ValueError:
                    continue
            if not steps:
                raise FileNotFoundError(f"No checkpoint named *_<step>.* in {cfg.DIR_CKPT} (found: {sorted(names)[:5]}).")
            ckpt = sorted(steps)[-1]
        if getattr(cfg, "TEST_EMA", False):
            cfg.DIR_CKPT = os.path.join(cfg.DIR_RESULT, "ema_ckpt")
        cfg.TEST_CKPT_PATH = os.path.join(cfg.DIR_CKPT, f"save_step_{ckpt}.pth")
        return str(ckpt) if not isinstance(ckpt, str) else ckpt, cfg.TEST_CKPT_PATH
    return "unknown", cfg.TEST_CKPT_PATH


def load_for_evaluation(net: torch.nn.Module, cfg, device=None):
    """select_checkpoint + load_network: what the evaluator does before it builds its engines.  Returns
    (net, ckpt label, removed keys)."""
    label, path = select_checkpoint(cfg)
    if path is None:
        return net, label, []
    net, removed = load_network(net, path, device)
    return net, label, removed
