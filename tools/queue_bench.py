"""Mixed-length dataset through the batched clip driver: slot queue (a finished slot takes the next clip,
BatchedClipDriver.run_queue) against lockstep batches (plan_ragged_batches + run_clips).  480p, K=4, B slots,
`--clips` synthetic clips with lengths drawn uniformly from [--min-frames, --max-frames]; one untimed pass of each
mode first (recordings, hipGraph captures), then one timed pass.  Prints one JSON line.

    python tools/queue_bench.py --clips 24 --slots 8 --min-frames 8 --max-frames 40
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=24)
    ap.add_argument("--slots", type=int, default=8)
    ap.add_argument("--min-frames", type=int, default=8)
    ap.add_argument("--max-frames", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--reproducible", action="store_true",
                    help="rmem_amd.determinism.reproducible_convolutions(): MIOpen without implicit GEMM, as the test "
                         "suite runs -- the per-clip hashes of the two modes are then expected to be EQUAL")
    a = ap.parse_args()
    if a.reproducible:
        from rmem_amd.determinism import reproducible_convolutions
        reproducible_convolutions()
    from rmem_amd import driver as D
    from rmem_amd.config import get_config
    from rmem_amd.model import build_vos_model
    from rmem_amd.synth import load_synthetic_weights, synth_clip
    dev = torch.device("cuda:0")
    H_IN, W_IN, H_OUT, W_OUT = 465, 833, 480, 854
    cfg = get_config("r50_deaotl", 1, 3)
    model = build_vos_model(cfg.MODEL_VOS, cfg).eval()
    load_synthetic_weights(model)
    model = model.to(dev)
    lens = [int(x) for x in np.random.RandomState(a.seed).randint(a.min_frames, a.max_frames + 1, size=a.clips)]

    def frames_of(cid, n):
        imgs, lab = synth_clip(cid, n, H_IN, W_IN, 3)
        lab0 = F.interpolate(lab, size=(H_OUT, W_OUT), mode="nearest").to(dev)
        return [D.make_samples(imgs[t].to(dev), lab0 if t == 0 else None, (H_OUT, W_OUT), 3, name=f"{t:05d}.jpg")
                for t in range(n)]
    clips = [frames_of(2000 + i, n) for i, n in enumerate(lens)]
    drv = D.BatchedClipDriver(model, a.slots, cfg)
    out = {"clips": a.clips, "slots": a.slots, "lengths": lens, "propagated_frames": sum(n - 1 for n in lens)}
    hashes = {}
    for mode in ("lockstep", "queue"):
        drv.run_dataset(clips, mode=mode)                      # untimed: recordings, graph captures
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = drv.run_dataset(clips, mode=mode)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        hashes[mode] = [hashlib.sha256(r.masks.cpu().numpy().tobytes()).hexdigest()[:16] for r in res]
        out[mode] = {"seconds": round(dt, 4), "frames_per_s": round(out["propagated_frames"] / dt, 1)}
        if mode == "queue":
            out[mode].update(drv.queue_stats)
        else:
            plan = D.plan_ragged_batches([drv.clip_info(c) for c in clips], a.slots, gap_of=drv._gap_of)
            out[mode].update({"steps": sum(max(lens[i] for i in b if i >= 0) for b in plan["batches"]),
                              "batches": len(plan["batches"]), "singles": len(plan["singles"])})
    out["reproducible_convolutions"] = bool(a.reproducible)
    out["clips_with_equal_masks"] = sum(x == y for x, y in zip(hashes["queue"], hashes["lockstep"]))
    out["speedup"] = round(out["queue"]["frames_per_s"] / out["lockstep"]["frames_per_s"], 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
