#!/usr/bin/env python
"""Closed loop (each engine is fed its OWN label maps) HIP engine vs CPU oracle on a small clip, for a
sweep of feedback gains: the ID-bank weights (the path by which a label map re-enters the memory)
scaled by s.  With the default synthetic weights the loop amplifies a single near-tie flip
(tests/test_oracle_golden.py); this probe looks for a gain at which the loop is stable AND the
memory still moves the masks (labels differ from a run whose memory never sees the labels).

    python tests/probes/closed_loop_probe.py [--frames 16] [--scales 1,0.5,0.25,0.1]
"""
import argparse, copy, os, sys
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def run(engine, imgs, lab, dev, H, W, frames, feed_zero=False):
    engine.restart_engine()
    engine.add_reference_frame(imgs[0].to(dev), lab.to(dev), obj_nums=[3], frame_step=0)
    out = []
    for t in range(1, frames):
        logit = engine.match_propogate_one_frame(imgs[t].to(dev), output_size=(H, W))
        pred = torch.argmax(logit, dim=1, keepdim=True).float()
        out.append(pred[0, 0].cpu().numpy().astype(np.uint8))
        fed = torch.zeros_like(pred) if feed_zero else pred
        engine.update_memory(F.interpolate(fed, size=engine.input_size_2d, mode="nearest"))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--scales", default="1,0.5,0.25,0.1")
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--hw", default="97,129")
    ap.add_argument("--gap", type=int, default=2)
    ap.add_argument("--no-zero-run", action="store_true")
    args = ap.parse_args()
    from oracle.engine_ref import OracleDeAOTInferEngine
    from rmem_amd.config import get_config
    from rmem_amd.engine import build_engine
    from rmem_amd.model import build_vos_model
    from rmem_amd.synth import load_synthetic_weights, synth_clip
    H, W = (int(x) for x in args.hw.split(","))
    imgs, lab = synth_clip(args.seed, args.frames, H, W, 3)
    for s in [float(x) for x in args.scales.split(",")]:
        cfg = get_config("r50_deaotl", 1, 3)
        cpu = build_vos_model("deaot", cfg).eval()
        load_synthetic_weights(cpu)
        with torch.no_grad():
            cpu.patch_wise_id_bank.weight.mul_(s)
            cpu.patch_wise_id_bank.bias.mul_(s)
        gpu = copy.deepcopy(cpu).to("cuda:0")
        ora = OracleDeAOTInferEngine(cpu, long_term_mem_gap=args.gap)
        hip_e = build_engine("deaotengine", phase="eval", aot_model=gpu, gpu_id=0, long_term_mem_gap=args.gap)
        hip_e.eval()
        a = run(ora, imgs, lab, "cpu", H, W, args.frames)
        b = run(hip_e, imgs, lab, "cuda:0", H, W, args.frames)
        z = a if args.no_zero_run else run(ora, imgs, lab, "cpu", H, W, args.frames, feed_zero=True)
        mism = [int((x != y).sum()) for x, y in zip(a, b)]
        moved = [int((x != y).sum()) for x, y in zip(a, z)]
        print(f"id scale {s}: closed-loop HIP vs oracle mismatching px/frame {mism}; memory moves (vs labels never fed) {moved}")


main()
