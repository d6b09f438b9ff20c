"""Clips with more than 10 objects (one sub-engine per 10 ids, engines/aot_engine.py:675-702): frames/s of the multi-object
wrapper with its sub-engines as slots of one batched engine (default) against one sub-engine after the other
(RMEM_MULTI_ENGINE=serial).  480p, K=4, steady-state bank, the clip driver's per-frame protocol with fused labels off
(generic path: logits -> argmax -> nearest resize -> update_memory).  Prints one JSON line.

    python tools/multiobj_bench.py --objects 12 23 --steps 60
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--objects", type=int, nargs="+", default=[12, 23])
    ap.add_argument("--steps", type=int, default=60)
    a = ap.parse_args()
    from inputs import multiobj_label
    from rmem_amd.config import get_config
    from rmem_amd.engine import build_engine
    from rmem_amd.model import build_vos_model
    from rmem_amd.synth import load_synthetic_weights, synth_clip
    dev = torch.device("cuda:0")
    H, W = 465, 833
    cfg = get_config("r50_deaotl", 1, 3)
    model = build_vos_model(cfg.MODEL_VOS, cfg).eval()
    load_synthetic_weights(model)
    model = model.to(dev)
    imgs, _ = synth_clip(7, 8, H, W, 3)
    imgs = [x.to(dev) for x in imgs]
    out = {"geometry": "480p (%dx%d), K=4, gap 5" % (H, W), "steps": a.steps}
    for n in a.objects:
        lab = multiobj_label(H, W, n).to(dev)
        res = {}
        for mode in ("serial", "batched"):
            os.environ["RMEM_MULTI_ENGINE"] = mode
            eng = build_engine(cfg.MODEL_ENGINE, phase="eval", aot_model=model, gpu_id=0, long_term_mem_gap=5)
            eng.eval()
            eng.add_reference_frame(imgs[0], lab, obj_nums=[n], frame_step=0)

            def step(t):
                lg = eng.match_propogate_one_frame(imgs[t % 8], output_size=(H, W), next_img=imgs[(t + 1) % 8])
                cur = F.interpolate(torch.argmax(lg, dim=1, keepdim=True).float(), size=eng.input_size_2d, mode="nearest")
                eng.update_memory(cur)
            for t in range(1, 31):           # bank full, every (slot, depth) state recorded / captured
                step(t)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(31, 31 + a.steps):
                step(t)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res[mode] = round(a.steps / dt, 1)
            res["sub_engines"] = len(eng.aot_engines)
            del eng
        res["speedup"] = round(res["batched"] / res["serial"], 2)
        out[f"{n}_objects"] = res
    os.environ.pop("RMEM_MULTI_ENGINE", None)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
