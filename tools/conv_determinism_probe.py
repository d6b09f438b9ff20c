#!/usr/bin/env python
"""Which convolution of the encoder / decoder is not bit-reproducible from call to call (same process, same
input), and which MIOpen solver family is responsible?

    python tools/conv_determinism_probe.py child     hooks every nn.Conv2d of the (folded) model, runs the same
                                                     481x849 frame 4 times, prints the modules whose output hash
                                                     changes between calls (name, weight shape, stride, input shape)
    python tools/conv_determinism_probe.py           parent: the child under several MIOPEN_DEBUG_* environments
"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MODES = {
    "default": {},
    "fix_random": {"RMEM_DETERMINISTIC": "1"},
    "no_implicit_gemm": {"MIOPEN_DEBUG_CONV_IMPLICIT_GEMM": "0"},
    "no_winograd": {"MIOPEN_DEBUG_CONV_WINOGRAD": "0"},
    "no_gemm": {"MIOPEN_DEBUG_CONV_GEMM": "0"},
    "no_direct": {"MIOPEN_DEBUG_CONV_DIRECT": "0"},
}


def child():
    import torch
    from rmem_amd.determinism import maybe_fix_random
    maybe_fix_random()
    from rmem_amd.config import get_config
    from rmem_amd.model import build_vos_model
    from rmem_amd.synth import load_synthetic_weights, synth_clip
    H, W = int(os.environ.get("PH", 481)), int(os.environ.get("PW", 849))
    dev = "cuda:0"
    cfg = get_config("r50_deaotl", 1, 3)
    model = build_vos_model(cfg.MODEL_VOS, cfg).eval()
    load_synthetic_weights(model)
    model = model.to(dev)
    model.optimize_for_inference(True)
    hh = lambda t: hashlib.sha256(t.detach().float().cpu().numpy().tobytes()).hexdigest()[:12]
    log = []
    import torch.nn.functional as F
    orig = F.conv2d

    def conv2d(x, w, *a, **k):                       # every convolution goes through here (nn.Conv2d included)
        out = orig(x, w, *a, **k)
        stride = a[1] if len(a) > 1 else k.get("stride", 1)
        log.append((f"conv{len(log)}", hh(out), tuple(w.shape), stride if isinstance(stride, tuple) else (stride,), tuple(x.shape)))
        return out
    F.conv2d = conv2d
    torch.nn.functional.conv2d = conv2d
    imgs, _ = synth_clip(7, 2, H, W, 3)
    x = imgs[1].to(dev)
    runs = []
    with torch.no_grad():
        for _ in range(4):
            del log[:]
            enc = model.encode_image(x)
            emb = torch.zeros(1, 512, enc[-1].shape[2], enc[-1].shape[3], device=dev)
            emb.normal_(generator=torch.Generator(device=dev).manual_seed(1))
            model.decoder([enc[-1], emb], enc)
            runs.append(list(log))
    bad = {}
    for r in runs[1:]:
        for a, b in zip(runs[0], r):
            if a[1] != b[1] and a[0] not in bad:
                bad[a[0]] = {"weight": a[2], "stride": a[3], "input": a[4]}
    first = next((a[0] for a, b in zip(runs[0], runs[1]) if a[1] != b[1]), None)
    print("PROBE " + json.dumps({"convs": len(runs[0]), "first_differing": first, "differing": bad}))


def parent():
    res = {}
    for mode, env in MODES.items():
        e = dict(os.environ)
        e.update(env)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=e, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("PROBE ")]
        res[mode] = json.loads(line[0][6:]) if line else {"error": (p.stderr or "")[-300:]}
        res[mode]["env"] = env
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    child() if len(sys.argv) > 1 and sys.argv[1] == "child" else parent()
