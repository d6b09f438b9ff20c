#!/usr/bin/env python
"""Attribute the mismatching label pixels of the golden 481x849 clip (teacher-forced, 9 frames of
409,920 pixels, gold = the reference's own label maps, tests/golden/clip_480p.npz) to the stages
of the frame: encoder (MIOpen, FrozenBN folding), LSTT (HIP kernels), decoder (MIOpen + fused
GroupNorm / skip merges).  Every row runs the same clip; stages are swapped between the GPU
product path and the CPU (PyTorch fp32) path of the same model:

  enc  lstt  dec
  gpu  hip   gpu    the product (row "product")
  cpu  hip   gpu    encoder error removed
  cpu  hip   cpu    LSTT error only           <- the number the north star asks to be 0
  cpu  cpu   gpu    decoder error only (oracle LSTT)
  cpu  cpu   cpu    the oracle itself (fp32 re-association only)

Environment switches are applied by the caller (one process per setting):
  RMEM_FOLD_BN=0                 FrozenBN not folded into the encoder convolutions
  MIOPEN_DEBUG_CONV_WINOGRAD=0   no Winograd solvers
  RMEM_P16=1                     bank reads with P as one fp16 plane

    python tests/probes/parity_attribution.py --tag base --out gpurun_out/parity_base.json
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="base")
    ap.add_argument("--out", default=None)
    ap.add_argument("--rows", default="product,cpuenc,lsttonly,deconly,oracle")
    args = ap.parse_args()
    rows = args.rows.split(",")
    from oracle.engine_ref import OracleDeAOTEngine
    from rmem_amd.config import get_config
    from rmem_amd.engine import DeAOTEngine, build_engine
    from rmem_amd.model import build_vos_model
    from rmem_amd.synth import load_synthetic_weights, synth_clip

    dev = "cuda:0"
    gd = os.path.join(ROOT, "tests", "golden")
    meta = json.load(open(os.path.join(gd, "clip_480p.json")))
    gold = np.load(os.path.join(gd, "clip_480p.npz"))["labels"]
    # the reference in double precision on the same teacher-forced inputs (make_golden.py:gen_clip_480p_fp64)
    gold64 = np.load(os.path.join(gd, "clip_480p_fp64.npz"))["labels64"]
    vs64 = {}
    cfg = get_config("r50_deaotl", meta["former"], meta["latter"])
    cpu_model = build_vos_model("deaot", cfg).eval()
    load_synthetic_weights(cpu_model)
    imgs, lab = synth_clip(meta["seed"], meta["frames"], meta["H"], meta["W"], 3)
    out_hw = tuple(meta["out_hw"])
    F_ = meta["frames"]
    fold = os.environ.get("RMEM_FOLD_BN", "1") == "1"

    def labels_of(logits):
        up = F.interpolate(logits, size=out_hw, mode="bilinear", align_corners=cfg.MODEL_ALIGN_CORNERS)
        return torch.argmax(up, dim=1)[0].cpu().numpy().astype(np.uint8)

    def fed(t, size):
        return F.interpolate(torch.from_numpy(gold[t - 1]).float()[None, None], size=size, mode="nearest")

    res = {"tag": args.tag, "env": {k: os.environ.get(k) for k in
                                   ("RMEM_FOLD_BN", "MIOPEN_DEBUG_CONV_WINOGRAD", "RMEM_P16", "RMEM_FUSED")},
           "pixels_per_frame": int(gold[0].size), "rows": {}}

    with torch.no_grad():
        # ---- CPU encoder pyramid of every frame, oracle run (teacher-forced) with its LSTT outputs
        enc_cpu = [cpu_model.encode_image(im) for im in imgs]
        ora_out, ora_lab = [], []
        if "oracle" in rows or "deconly" in rows:
            ora = OracleDeAOTEngine(cpu_model, long_term_mem_gap=meta["gap"])
            ora.add_reference_frame(imgs[0], lab, obj_nums=[3], frame_step=0)
            for t in range(1, F_):
                lg = ora.match_propogate_one_frame(imgs[t], output_size=None)
                ora_out.append(ora.last_lstt_out.clone())
                ora_lab.append(labels_of(lg))
                ora.update_memory(fed(t, ora.input_size_2d))
            res["rows"]["oracle (cpu/cpu/cpu)"] = [int((a != gold[i]).sum()) for i, a in enumerate(ora_lab)]
            vs64["oracle (cpu/cpu/cpu)"] = [int((a != gold64[i]).sum()) for i, a in enumerate(ora_lab)]

        gpu_model = copy.deepcopy(cpu_model).to(dev)
        # ---- product
        if "product" in rows:
            eng = build_engine(cfg.MODEL_ENGINE, phase="eval", aot_model=gpu_model, gpu_id=0,
                               long_term_mem_gap=meta["gap"], fold_bn=fold)
            eng.eval()
            gi = [x.to(dev) for x in imgs]
            eng.add_reference_frame(gi[0], lab.to(dev), obj_nums=[3], frame_step=0)
            mm, m64 = [], []
            for t in range(1, F_):
                lg = eng.match_propogate_one_frame(gi[t], output_size=None)
                lb = labels_of(lg)
                mm.append(int((lb != gold[t - 1]).sum()))
                m64.append(int((lb != gold64[t - 1]).sum()))
                eng.update_memory(fed(t, eng.input_size_2d).to(dev))
            res["rows"]["product (gpu/hip/gpu)"] = mm
            vs64["product (gpu/hip/gpu)"] = m64
            res["indexes_ok"] = list(eng.aot_engines[0].long_memories_indexes) == meta["indexes"][-1]
        # ---- CPU encoder features -> HIP LSTT -> GPU decoder / CPU decoder
        if "cpuenc" in rows or "lsttonly" in rows:
            plain = copy.deepcopy(cpu_model).to(dev)          # no folded encoder needed: features are given
            sub = DeAOTEngine(plain, 0, long_term_mem_gap=meta["gap"], use_graphs=False)
            sub.eval()
            eg = [[x.to(dev) for x in e] for e in enc_cpu]
            sub.add_reference_frame(imgs[0].to(dev), lab.to(dev), obj_nums=[10], img_embs=eg[0], frame_step=0)
            mm_g, mm_c, lerr, g64, c64 = [], [], [], [], []
            for t in range(1, F_):
                lg = sub.match_propogate_one_frame(img=None, img_embs=eg[t], output_size=None)
                lb = labels_of(lg)
                mm_g.append(int((lb != gold[t - 1]).sum()))
                g64.append(int((lb != gold64[t - 1]).sum()))
                out = sub.lstt.out.cpu()
                lc = cpu_model.decode_id_logits(out, enc_cpu[t])
                lb = labels_of(lc)
                mm_c.append(int((lb != gold[t - 1]).sum()))
                c64.append(int((lb != gold64[t - 1]).sum()))
                if ora_out:
                    lerr.append(float((out - ora_out[t - 1]).abs().max()))
                sub.update_short_term_memory(fed(t, sub.input_size_2d).to(dev))
            res["rows"]["cpu enc -> HIP LSTT -> gpu dec"] = mm_g
            res["rows"]["cpu enc -> HIP LSTT -> cpu dec (LSTT only)"] = mm_c
            res["lstt_out_max_abs_err_vs_oracle"] = lerr
            vs64["cpu enc -> HIP LSTT -> gpu dec"] = g64
            vs64["cpu enc -> HIP LSTT -> cpu dec (LSTT only)"] = c64
        # ---- oracle LSTT output -> GPU decoder
        if "deconly" in rows and ora_out:
            mm, m64 = [], []
            for t in range(1, F_):
                eg = [x.to(dev) for x in enc_cpu[t]]
                lg = gpu_model.decode_id_logits(ora_out[t - 1].to(dev), eg)
                lb = labels_of(lg)
                mm.append(int((lb != gold[t - 1]).sum()))
                m64.append(int((lb != gold64[t - 1]).sum()))
            res["rows"]["cpu enc -> oracle LSTT -> gpu dec (decoder only)"] = mm
            vs64["cpu enc -> oracle LSTT -> gpu dec (decoder only)"] = m64
    res["sums"] = {k: int(sum(v)) for k, v in res["rows"].items()}
    res["rows_vs_fp64_reference"] = vs64
    res["sums_vs_fp64_reference"] = {k: int(sum(v)) for k, v in vs64.items()}
    res["fp32_reference_vs_fp64_reference"] = [int((gold[i] != gold64[i]).sum()) for i in range(F_ - 1)]
    print(json.dumps(res))
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
