"""Import harness for the *reference* implementation (this container only).

Used by ``make_golden.py`` (fixture generation) and by
``tests/test_oracle_vs_reference.py`` (skipped when /root/reference is absent, e.g.
on the GPU box).  Follows the recipe of SURVEY.md Appendix D: stub the modules
the image lacks (timm, torchvision), never import managers/dataloaders, run from a
scratch cwd (configs create ./results), never write bytecode into the reference
tree, and give ``AOTEngine.assign_identity`` an explicit CPU ignore-mask instead of
its hard-coded CUDA device (aot_engine.py:209-213).

Nothing here is copied from the reference: it only imports it.
"""
from __future__ import annotations

import contextlib
import io
import os
import sys
import tempfile
import types

REF_ROOT = "/root/reference/aot_plus"


def available() -> bool:
    return os.path.isdir(REF_ROOT)


def _stub_modules():
    import torch
    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")
        models = types.ModuleType("timm.models")
        layers = types.ModuleType("timm.models.layers")
        layers.trunc_normal_ = torch.nn.init.trunc_normal_
        layers.DropPath = torch.nn.Identity
        layers.to_2tuple = lambda x: (x, x) if not isinstance(x, tuple) else x
        timm.models = models
        models.layers = layers
        sys.modules.update({"timm": timm, "timm.models": models,
                            "timm.models.layers": layers})
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tr = types.ModuleType("torchvision.transforms")
        fn = types.ModuleType("torchvision.transforms.functional")

        class InterpolationMode:  # noqa: D401 - attribute bag
            NEAREST = "nearest"
            BILINEAR = "bilinear"

        tr.InterpolationMode = InterpolationMode
        tr.functional = fn
        tv.transforms = tr
        sys.modules.update({"torchvision": tv, "torchvision.transforms": tr,
                            "torchvision.transforms.functional": fn})


_IMPORTED = {}


def import_reference():
    """Returns dict(get_config, build_vos_model, build_engine, modules...)."""
    if _IMPORTED:
        return _IMPORTED
    if not available():
        raise RuntimeError("reference tree not present")
    sys.dont_write_bytecode = True
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    _stub_modules()
    scratch = tempfile.mkdtemp(prefix="rmem_ref_")
    os.chdir(scratch)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    with contextlib.redirect_stdout(io.StringIO()):
        from tools.get_config import get_config
        from networks.models import build_vos_model
        from networks.engines import build_engine
        from networks.engines import aot_engine as aot_engine_mod
        from networks.layers import attention as attention_mod
        from networks.layers import transformer as transformer_mod
    import torch

    orig_assign = aot_engine_mod.AOTEngine.assign_identity

    def assign_identity_cpu(self, one_hot_mask, ignore_mask=None):
        if ignore_mask is None:
            ignore_mask = torch.zeros(one_hot_mask.shape[0], 1, one_hot_mask.shape[2],
                                      one_hot_mask.shape[3], device=one_hot_mask.device)
        return orig_assign(self, one_hot_mask, ignore_mask)

    aot_engine_mod.AOTEngine.assign_identity = assign_identity_cpu
    _IMPORTED.update(get_config=get_config, build_vos_model=build_vos_model,
                     build_engine=build_engine, aot_engine=aot_engine_mod,
                     attention=attention_mod, transformer=transformer_mod)
    return _IMPORTED


def build_reference(model_name="r50_deaotl", former=1, latter=3, gap=5, salt=0):
    """Reference model + engine on CPU with the name-keyed synthetic weights."""
    import torch
    ref = import_reference()
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = ref["get_config"]("pre_vost", "golden", model_name)
    if model_name == "swinb_aotl":
        # the shipped swinb config lacks the RMem attributes (AOT.__init__ raises at aot.py:23);
        # inject the r50_aotl values, as SURVEY.md section 7 prescribes
        import importlib
        donor = importlib.import_module("configs.models.r50_aotl").ModelConfig()
        for k, v in donor.__dict__.items():
            if not hasattr(cfg, k):
                setattr(cfg, k, v)
    cfg.FORMER_MEM_LEN = former
    cfg.LATTER_MEM_LEN = latter
    with contextlib.redirect_stdout(io.StringIO()):
        model = ref["build_vos_model"](cfg.MODEL_VOS, cfg).eval()
    from rmem_amd.synth import load_synthetic_weights
    load_synthetic_weights(model, salt)
    engine = ref["build_engine"](cfg.MODEL_ENGINE, phase="eval", aot_model=model,
                                 gpu_id=0, long_term_mem_gap=gap)
    engine.eval()
    return cfg, model, engine


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield
