#!/usr/bin/env python
"""The folded encoder gives different bytes from call to call WITHOUT a host sync between its kernels, and
identical bytes WITH one after every convolution (tools/conv_determinism_probe.py).  Where does it start?
Every module output is cloned on the stream (no host sync inside a pass) and hashed after the pass; prints the
first module whose output differs between passes, for the default launch mode and with AMD_SERIALIZE_KERNEL=3."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    from rmem_amd.config import get_config
    from rmem_amd.model import build_vos_model
    from rmem_amd.synth import load_synthetic_weights, synth_clip
    dev = "cuda:0"
    cfg = get_config("r50_deaotl", 1, 3)
    model = build_vos_model(cfg.MODEL_VOS, cfg).eval()
    load_synthetic_weights(model)
    model = model.to(dev)
    fold = os.environ.get("PROBE_FOLD", "1") == "1"
    if fold:
        model.optimize_for_inference(True)
    enc = model.__dict__.get("_enc_infer") or model.encoder
    whole = os.environ.get("PROBE_WHOLE", "0") == "1"        # hook model.encode_image's extras too (projector, FPN adapters)
    hh = lambda t: hashlib.sha256(t.detach().float().cpu().numpy().tobytes()).hexdigest()[:12]
    kept = []
    mods = list(enc.named_modules())
    if whole:
        mods += [("model." + n, m) for n, m in model.named_modules() if n and not n.startswith("encoder.")]
    for name, mod in mods:
        if name == "":
            continue
        mod.register_forward_hook(lambda m, i, o, name=name: kept.append((name, type(m).__name__, o.clone() if torch.is_tensor(o) else None)))
    imgs, _ = synth_clip(7, 2, int(os.environ.get("PH", 481)), int(os.environ.get("PW", 849)), 3)
    x = imgs[1].to(dev)
    runs = []
    with torch.no_grad():
        for _ in range(4):
            del kept[:]
            if whole:
                model.encode_image(x)
            else:
                enc(x)
            torch.cuda.synchronize()
            runs.append([(n, t, hh(o)) for n, t, o in kept if o is not None])
    out = {"fold": fold, "modules": len(runs[0])}
    for r in (1, 2, 3):
        diff = [(a[0], a[1]) for a, b in zip(runs[0], runs[r]) if a[2] != b[2]]
        out[f"pass{r + 1}_vs_1_first_differing"] = diff[:3]
        out[f"pass{r + 1}_vs_1_count"] = len(diff)
    print("PROBE " + json.dumps(out))


def parent():
    res = {}
    for mode, env in (("default", {"PROBE_WHOLE": "1"}), ("serialize_kernels", {"AMD_SERIALIZE_KERNEL": "3", "PROBE_WHOLE": "1"}),
                      ("unfolded_bn", {"PROBE_FOLD": "0", "PROBE_WHOLE": "1"}),
                      ("no_winograd", {"MIOPEN_DEBUG_CONV_WINOGRAD": "0", "PROBE_WHOLE": "1"}),
                      ("no_implicit_gemm", {"MIOPEN_DEBUG_CONV_IMPLICIT_GEMM": "0", "PROBE_WHOLE": "1"}),
                      ("no_direct", {"MIOPEN_DEBUG_CONV_DIRECT": "0", "PROBE_WHOLE": "1"}),
                      ("no_gemm", {"MIOPEN_DEBUG_CONV_GEMM": "0", "PROBE_WHOLE": "1"})):
        e = dict(os.environ)
        e.update(env)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=e, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("PROBE ")]
        res[mode] = json.loads(line[0][6:]) if line else {"error": (p.stderr or "")[-300:]}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    child() if len(sys.argv) > 1 and sys.argv[1] == "child" else parent()
