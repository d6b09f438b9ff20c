#!/usr/bin/env python
"""Are the MIOpen stages bit-reproducible BETWEEN PROCESSES, and does the reference's --fix_random
switch (tools/eval.py:21-37 -> rmem_amd.determinism.fix_random) change that?

    python tools/parity_mode_probe.py child <mode>     one process: sha256 of the encoder pyramid, the LSTT
                                                       output and the decoder logits of 3 frames, as JSON
    python tools/parity_mode_probe.py                  parent: runs every mode in TWO fresh processes and
                                                       reports which stages hash equal

Modes: default | det (fix_random) | det_nowino (+ MIOPEN_DEBUG_CONV_WINOGRAD=0) | det_nogemm
(+ MIOPEN_DEBUG_CONV_GEMM=0) | det_direct (only direct / implicit-GEMM solvers: Winograd, GEMM and FFT off).
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
    "det": {"RMEM_DETERMINISTIC": "1"},
    "det_nowino": {"RMEM_DETERMINISTIC": "1", "MIOPEN_DEBUG_CONV_WINOGRAD": "0"},
    "det_nogemm": {"RMEM_DETERMINISTIC": "1", "MIOPEN_DEBUG_CONV_GEMM": "0"},
    "det_direct": {"RMEM_DETERMINISTIC": "1", "MIOPEN_DEBUG_CONV_WINOGRAD": "0", "MIOPEN_DEBUG_CONV_GEMM": "0",
                   "MIOPEN_DEBUG_CONV_FFT": "0"},
}


def child():
    import torch
    from rmem_amd.determinism import maybe_fix_random
    maybe_fix_random()
    from rmem_amd.config import get_config
    from rmem_amd.engine import build_engine
    from rmem_amd.model import build_vos_model
    from rmem_amd.synth import load_synthetic_weights, synth_clip
    H, W = int(os.environ.get("PH", 481)), int(os.environ.get("PW", 849))
    dev = "cuda:0"
    cfg = get_config("r50_deaotl", 1, 3)
    model = build_vos_model(cfg.MODEL_VOS, cfg).eval()
    load_synthetic_weights(model)
    model = model.to(dev)
    eng = build_engine(cfg.MODEL_ENGINE, phase="eval", aot_model=model, gpu_id=0, long_term_mem_gap=2)
    eng.eval()
    imgs, lab = synth_clip(7, 4, H, W, 3)
    hh = lambda t: hashlib.sha256(t.detach().float().cpu().numpy().tobytes()).hexdigest()[:12]
    out = {"enc": [], "lstt": [], "logits": []}
    with torch.no_grad():
        for t in range(1, 4):                       # the (folded) encoder alone, batch 1 and batch 2
            out["enc"].append([hh(f) for f in model.encode_image(imgs[t].to(dev))])
        out["enc_b2"] = [hh(f) for f in model.encode_image(torch.cat([imgs[1], imgs[2]]).to(dev))]
        eng.add_reference_frame(imgs[0].to(dev), lab.to(dev), obj_nums=[3], frame_step=0)
        for t in range(1, 4):
            lg = eng.match_propogate_one_frame(imgs[t].to(dev), output_size=(H, W))
            sub = eng.aot_engines[0]
            out["lstt"].append(hh(sub.lstt.out))
            out["logits"].append(hh(sub.pred_id_logits))
            eng.update_memory(torch.nn.functional.interpolate(lg.argmax(1, keepdim=True).float(), size=eng.input_size_2d,
                                                              mode="nearest"))
    print("PROBE " + json.dumps(out))


def parent():
    res = {}
    for mode, env in MODES.items():
        runs = []
        for _ in range(2):
            e = dict(os.environ)
            e.update(env)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", mode], env=e, capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("PROBE ")]
            if not line:
                runs.append({"error": (p.stderr or "")[-400:]})
                continue
            runs.append(json.loads(line[0][6:]))
        if any("error" in r for r in runs):
            res[mode] = {"error": [r.get("error") for r in runs]}
            continue
        a, b = runs
        res[mode] = {"env": env,
                     "encoder_equal_per_frame": [x == y for x, y in zip(a["enc"], b["enc"])],
                     "encoder_levels_equal_frame1": [x == y for x, y in zip(a["enc"][0], b["enc"][0])],
                     "encoder_batch2_equal": a["enc_b2"] == b["enc_b2"],
                     "lstt_equal_per_frame": [x == y for x, y in zip(a["lstt"], b["lstt"])],
                     "decoder_logits_equal_per_frame": [x == y for x, y in zip(a["logits"], b["logits"])]}
    print(json.dumps({"size": [int(os.environ.get("PH", 481)), int(os.environ.get("PW", 849))], "modes": res}, indent=1))


if __name__ == "__main__":
    child() if len(sys.argv) > 1 and sys.argv[1] == "child" else parent()
