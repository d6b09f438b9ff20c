"""Generate the golden fixtures in this directory by running the *reference*
(/root/reference, imported on CPU through refharness.py) on seeded synthetic inputs
with the name-keyed synthetic weights of ``rmem_amd.synth``.

Run (build container only; the reference never travels):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Fixtures are data only (inputs + the reference's outputs):
  manifest_r50_deaotl.json   state_dict keys / shapes
  block_*.npz                one GatedPropagationModule.forward call (transformer.py:1091)
  idassign_*.npz             one_hot_mask + assign_identity + get_id_emb
  clip_small_*.json/.npz     engine state machine on small clips (indexes, EMA, visits, labels)
  clip_480p.json/.npz        481x849 clip: per-frame label hashes + a few logits (fp16)
  multiengine_wrapper.*      AOTInferEngine.separate_mask / soft_logit_aggregation (> 10 objects)
  clip_480p_fp64.*           the same 481x849 clip through the reference in DOUBLE precision (near-tie arbitration)
  clip_tta_ms_gap2.*         multi-scale x flip test-time augmentation (four engines, two image sizes), new object mid-clip
  clip_480p_long*.json/.npz  the 481x849 clip at the evaluator's gap 5 over 46 frames (six evictions) + its fp64 tie lists
  clip_720p_k8*.json/.npz    721x1281, K = 8, gap 1, 11 frames (bank full, eviction) + its fp64 tie lists
  clip_aot_480p_fp64.*, clip_swin_k4_gap2_fp64.*, clip_swin_480p*   fp64 tie lists of the AOT / Swin clips; SwinB-AOTL at 480x848
  load_network_cases.*       the reference's load_network (utils/checkpoint.py:75-101) over ten payload variants
  *_amp.json/.npz            golden clips through the reference under fp16 autocast (its --amp mode), teacher-forced
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

import refharness as rh  # noqa: E402
from rmem_amd.synth import synth_clip  # noqa: E402
from make_golden_inputs import ignore_region_label, tta_new_object_label, tta_scaled_images  # noqa: E402
from inputs import (AOT_BLOCK_CASES, BLOCK_CASES, IDASSIGN_CASES, aot_block_case_name,  # noqa: E402
                    aot_block_inputs, block_case_name, block_inputs, idassign_label)


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.to(torch.uint8).contiguous().numpy().tobytes()).hexdigest()


def to2d(x_nc, h, w):
    """[N,C] -> [1,C,h,w] (seq_to_2d, layers/basic.py:73-77)."""
    return x_nc.view(h, w, 1, -1).permute(2, 3, 0, 1).contiguous()


def gen_manifest(model):
    man = {k: list(v.shape) for k, v in model.state_dict().items()}
    with open(os.path.join(HERE, "manifest_r50_deaotl.json"), "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)


def gen_blocks(model):
    """Direct calls of the reference block on seeded inputs (tests/golden/inputs.py)."""
    temporal = torch.cat((model.cur_pos_emb, model.mem_pos_emb), dim=0)
    for layer, T, h, w, ref_frame in BLOCK_CASES:
        blk = model.LSTT.layers[layer]
        blk.short_term_attn.qk_mask = None
        blk.short_term_attn.local_mask = None
        blk.short_term_attn.last_size_2d = None
        i = block_inputs(layer, T, h, w, ref_frame)
        tgt, tgt_id = i["tgt"], i["tgt_id"]
        kw = dict(size_2d=(h, w), temporal_encoding=temporal, save_atten_weights=not ref_frame)
        # the pre-softmax logits of the long-term GatedPropagation (attention.py:184-187: QK = (Q / T) @ K, then
        # torch.softmax(QK, dim=-1)) and of the windowed read (attention.py:344-346: softmax over the 225 offsets): the
        # inputs of the block's softmax calls are recorded (nothing is changed), told apart by their shapes
        seen, orig_softmax = [], torch.softmax

        def recording_softmax(x, dim, **kws):
            seen.append((x.detach().clone(), dim))
            return orig_softmax(x, dim, **kws)

        torch.softmax = recording_softmax
        try:
            with torch.no_grad(), rh.quiet():
                if ref_frame:
                    out = blk(tgt[:, None], None if tgt_id is None else tgt_id[:, None],
                              None, None, curr_id_emb=i["id_emb"][:, None], **kw)
                else:
                    out = blk(tgt[:, None], None if tgt_id is None else tgt_id[:, None],
                              [i["bank_K"][:, :, None], i["bank_V"][:, :, None], None,
                               i["bank_IDV"][:, :, None]],
                              [to2d(i["short_K"], h, w), to2d(i["short_V"], h, w), None,
                               to2d(i["short_IDV"], h, w)], **kw)
        finally:
            torch.softmax = orig_softmax
        n = h * w
        Tm = 1 if ref_frame else T
        lt = [x for x, dim in seen if x.dim() == 4 and x.shape[-2] == n and x.shape[-1] == Tm * n and dim in (-1, 3)]
        st = [x for x, dim in seen if x.dim() == 4 and x.shape[-2] == 225 and x.shape[-1] == n]
        assert (len(lt) >= 1 or ref_frame) and len(st) == 1, [tuple(x.shape) for x, _ in seen]   # (reference frames take the SDPA branch: save_atten_weights is off)
        o_tgt, o_id, mems = out
        curr, glob, loc = mems
        d = dict(out_tgt=o_tgt[:, 0].numpy(), out_tgt_id=o_id[:, 0].numpy(),
                 curr_K=curr[0][:, 0].numpy(), curr_V=curr[1][:, 0].numpy(),
                 curr_z=np.zeros((0,), np.float32) if curr[3] is None else curr[3][:, 0].numpy())
        if ref_frame:
            d.update(glob_IDV=glob[3][0, :, 0].numpy())
        else:
            d.update(mass=blk.record_attn_weight.numpy())
        # [N][T*N] long-term logits (the first softmax of that shape: with T = 1 a second one cannot occur -- the self
        # attention goes through scaled_dot_product_attention), [225][N] windowed logits (-1e8 outside the image)
        d.update(st_logits=st[0][0, 0].numpy())
        if lt:
            d.update(lt_logits=lt[0][0, 0].numpy())
        np.savez_compressed(os.path.join(HERE, block_case_name(layer, T, h, w, ref_frame) + ".npz"), **d)


def gen_block_fullsize(model, layer=1, T=4, h=31, w=54, nrows=48):
    """One block call of the reference at BASELINE.json's 480p K = 4 size (31 x 54 = 1674 tokens, four bank slots) for the
    per-logit check at a benchmarked size: the inputs are inputs.block_inputs' seeded tensors; kept are the block's Q
    projection (curr_K, the read's query operand) and, for `nrows` query rows spread over the image (first / last rows and
    columns included), the reference's pre-softmax logits -- [rows][T * N] of the long-term read (attention.py:184-187) and
    [225][rows] of the windowed read (attention.py:344-346) -- recorded from its own softmax calls as gen_blocks does."""
    temporal = torch.cat((model.cur_pos_emb, model.mem_pos_emb), dim=0)
    blk = model.LSTT.layers[layer]
    blk.short_term_attn.qk_mask = None
    blk.short_term_attn.local_mask = None
    blk.short_term_attn.last_size_2d = None
    i = block_inputs(layer, T, h, w, False)
    seen, orig_softmax = [], torch.softmax

    def recording_softmax(x, dim, **kws):
        seen.append((x.detach().clone(), dim))
        return orig_softmax(x, dim, **kws)

    torch.softmax = recording_softmax
    try:
        with torch.no_grad(), rh.quiet():
            out = blk(i["tgt"][:, None], i["tgt_id"][:, None],
                      [i["bank_K"][:, :, None], i["bank_V"][:, :, None], None, i["bank_IDV"][:, :, None]],
                      [to2d(i["short_K"], h, w), to2d(i["short_V"], h, w), None, to2d(i["short_IDV"], h, w)],
                      size_2d=(h, w), temporal_encoding=temporal, save_atten_weights=True)
    finally:
        torch.softmax = orig_softmax
    n = h * w
    lt = [x for x, dim in seen if x.dim() == 4 and x.shape[-2] == n and x.shape[-1] == T * n and dim in (-1, 3)]
    st = [x for x, dim in seen if x.dim() == 4 and x.shape[-2] == 225 and x.shape[-1] == n]
    assert len(lt) >= 1 and len(st) == 1, [tuple(x.shape) for x, _ in seen]
    rs = np.random.RandomState(31 * 54)
    rows = sorted(set([0, w - 1, n - w, n - 1, 7 * w + 7, (h // 2) * w + w // 2] + rs.choice(n, nrows, replace=False).tolist()))[:nrows]
    rows = np.array(rows, dtype=np.int32)
    curr = out[2][0]
    np.savez_compressed(os.path.join(HERE, f"block_l{layer}_T{T}_{h}x{w}_rows.npz"), rows=rows, curr_K=curr[0][:, 0].numpy(),
                        lt_logits_rows=lt[0][0, 0][torch.from_numpy(rows).long()].numpy(),
                        st_logits_rows=st[0][0, 0][:, torch.from_numpy(rows).long()].numpy())
    print("full-size block fixture:", len(rows), "rows; |lt logit| up to", float(lt[0].abs().max()))


def gen_idassign(model):
    ref = rh.import_reference()
    from utils.image import one_hot_mask            # reference util (import only)
    eng = ref["aot_engine"].AOTEngine(model, 0, 5)
    for (H, W) in IDASSIGN_CASES:
        label = idassign_label(H, W)
        eng.restart_engine()
        eh, ew = (H - 1) // 16 + 1, (W - 1) // 16 + 1
        eng.update_size((H, W), (eh, ew))
        with torch.no_grad():
            oh, ign = one_hot_mask(label, 10)
            emb = eng.assign_identity(oh, ign)      # [N,1,256]
        np.savez_compressed(os.path.join(HERE, f"idassign_{H}x{W}.npz"),
                            id_emb=emb[:, 0].numpy(), eh=eh, ew=ew)


def gen_aot_blocks(model):
    """Direct calls of the reference AOT block (SimplifiedTransformerBlock.forward)."""
    from oracle.aot_ref import sine_pos_emb
    temporal = torch.cat((model.cur_pos_emb, model.mem_pos_emb), dim=0)
    for layer, T, h, w, ref_frame in AOT_BLOCK_CASES:
        blk = model.LSTT.layers[layer]
        i = aot_block_inputs(layer, T, h, w, ref_frame)
        with torch.no_grad():
            pos_ref = model.get_pos_emb(torch.zeros(1, 256, h, w)).view(1, -1, h * w).permute(2, 0, 1)
        assert (pos_ref[:, 0] - sine_pos_emb(h, w)).abs().max() < 1e-6
        kw = dict(self_pos=pos_ref, size_2d=(h, w), temporal_encoding=temporal,
                  save_atten_weights=not ref_frame)
        with torch.no_grad(), rh.quiet():
            if ref_frame:
                out = blk(i["tgt"][:, None], None, None, curr_id_emb=i["id_emb"][:, None], **kw)
            else:
                out = blk(i["tgt"][:, None], [i["bank_K"][:, :, None], i["bank_V"][:, :, None]],
                          [i["short_K"][:, None], i["short_V"][:, None]], **kw)
        o_tgt, mems = out
        curr, glob, loc = mems
        d = dict(out_tgt=o_tgt[:, 0].numpy(), curr_K=curr[0][:, 0].numpy(), curr_V=curr[1][:, 0].numpy(),
                 local_K=loc[0][:, 0].numpy(), local_V=loc[1][:, 0].numpy(), pos=pos_ref[:, 0].numpy())
        if ref_frame:
            d.update(glob_V=glob[1][0, :, 0].numpy())
        else:
            d.update(mass=blk.record_attn_weight.numpy())
        np.savez_compressed(os.path.join(HERE, aot_block_case_name(layer, T, h, w, ref_frame) + ".npz"), **d)


def gen_aot_clips():
    """BASELINE.json configs[0]: R50-AOTL + RMem, one synthetic 480p clip x 16 frames, K=4
    (gap 5 = evaluator rule, no eviction in 16 frames) plus small clips that do evict."""
    small = [("aot_k4_gap2", 97, 129, 16, 2, 1, 3), ("aot_k2_gap1", 81, 97, 10, 1, 1, 1)]
    for name, H, W, frames, gap, former, latter in small:
        cfg, model, engine = rh.build_reference("r50_aotl", former, latter, gap)
        imgs, lab = synth_clip(11, frames, H, W, 3)
        rec = run_reference_clip(engine, imgs, lab, (H, W), capture_logits=(frames - 1,))
        meta = dict(H=H, W=W, frames=frames, gap=gap, former=former, latter=latter, seed=11,
                    indexes=rec["indexes"], ema=rec["ema"], visits=rec["visits"], hist=rec["hist"],
                    label_sha=[sha(l) for l in rec["labels"]])
        json.dump(meta, open(os.path.join(HERE, f"clip_{name}.json"), "w"))
        np.savez_compressed(os.path.join(HERE, f"clip_{name}.npz"), labels=torch.stack(rec["labels"]).numpy(),
                            last_logits=rec["logits"][frames - 1].numpy())
        print("clip", name, "indexes", rec["indexes"][-1])
    cfg, model, engine = rh.build_reference("r50_aotl", 1, 3, 5)
    H, W, frames = 481, 849, 16
    imgs, lab = synth_clip(0, frames, H, W, 3)
    rec = run_reference_clip(engine, imgs, lab, (480, 854), capture_logits=(1, 15))
    meta = dict(H=H, W=W, out_hw=[480, 854], frames=frames, gap=5, former=1, latter=3, seed=0,
                indexes=rec["indexes"], hist=rec["hist"], label_sha=[sha(l) for l in rec["labels"]])
    json.dump(meta, open(os.path.join(HERE, "clip_aot_480p.json"), "w"))
    np.savez_compressed(os.path.join(HERE, "clip_aot_480p.npz"),
                        **{f"logits_{t}": v.numpy().astype(np.float16) for t, v in rec["logits"].items()},
                        labels=torch.stack(rec["labels"]).numpy())
    print("clip aot 480p indexes", rec["indexes"][-1])


# NOTE (>10 objects): the reference's multi-engine path cannot produce a golden vector -- all
# sub-engines of AOTInferEngine share one model and therefore ONE LSTT memory state
# (layers/transformer.py:1000-1007 lives on the shared module), so the second engine's
# update_short_memories re-fuses already fused memories and raises a shape error
# (transformer.py:1241: 63x768 vs 256x512) for 12 objects.  rmem_amd keeps the state per
# engine; tests/test_hip_engine.py::test_multi_object_engines checks it by property instead.


def gen_swin():
    """BASELINE.json configs[4] (SwinB-AOTL + RMem attributes injected): encoder features and a
    small clip through the reference engine."""
    cfg, model, engine = rh.build_reference("swinb_aotl", 1, 3, 2)
    json.dump({k: list(v.shape) for k, v in model.state_dict().items()},
              open(os.path.join(HERE, "manifest_swinb_aotl.json"), "w"), indent=0, sort_keys=True)
    H, W, frames = 128, 160, 12
    imgs, lab = synth_clip(7, frames, H, W, 3)
    with torch.no_grad():
        feats = model.encoder(imgs[0])
    rec = run_reference_clip(engine, imgs, lab, (H, W), capture_logits=(frames - 1,))
    meta = dict(H=H, W=W, frames=frames, gap=2, former=1, latter=3, seed=7, indexes=rec["indexes"],
                hist=rec["hist"])
    json.dump(meta, open(os.path.join(HERE, "clip_swin_k4_gap2.json"), "w"))
    np.savez_compressed(os.path.join(HERE, "clip_swin_k4_gap2.npz"), labels=torch.stack(rec["labels"]).numpy(),
                        last_logits=rec["logits"][frames - 1].numpy(),
                        enc16=feats[2].numpy().astype(np.float16), enc4_mean=feats[0].mean(dim=(2, 3)).numpy())
    print("swin clip indexes", rec["indexes"][-1], "labels", sorted(set(torch.stack(rec["labels"]).flatten().tolist())))


def run_reference_clip(engine, imgs, label0, out_hw, capture_logits=()):
    """Drives the reference engine with the evaluator's protocol
    (managers/evaluator.py:384-441,518-523) and records state after every frame."""
    sub = None
    rec = dict(indexes=[], ema=[], visits=[], labels=[], hist=[], logits={}, mass0=[])
    with torch.no_grad(), rh.quiet():
        engine.restart_engine()
        engine.add_reference_frame(imgs[0], label0.int(), obj_nums=[int(label0.max())], frame_step=0)
        sub = engine.aot_engines[0]
        lstt = sub.AOT.LSTT
        for t in range(1, len(imgs)):
            logit = engine.match_propogate_one_frame(imgs[t], output_size=out_hw)
            prob = torch.softmax(logit, dim=1)
            pred = torch.argmax(prob, dim=1, keepdim=True).float()
            cur = F.interpolate(pred, size=engine.input_size_2d, mode="nearest")
            engine.update_memory(cur)
            rec["indexes"].append(list(sub.long_memories_indexes))
            rec["ema"].append({int(k): float(v) for k, v in lstt.stored_attn_weight_dict.items()})
            rec["visits"].append({int(k): int(v) for k, v in lstt.stored_frame_times.items()})
            rec["labels"].append(pred[0, 0].to(torch.uint8))
            rec["hist"].append(torch.bincount(pred.flatten().long(), minlength=11).tolist())
            rec["mass0"].append(lstt.layers[0].record_attn_weight.sum(0).tolist())
            if t in capture_logits:
                rec["logits"][t] = sub.pred_id_logits.clone()
    return rec


def gen_clips():
    small = [  # name, H, W, frames, gap, former, latter
        ("k4_gap2", 97, 129, 16, 2, 1, 3),
        ("k4_gap5", 97, 129, 24, 5, 1, 3),
        ("k8_gap2", 81, 113, 28, 2, 1, 7),
        ("k2_gap1", 81, 97, 10, 1, 1, 1),
    ]
    for name, H, W, frames, gap, former, latter in small:
        cfg, model, engine = rh.build_reference("r50_deaotl", former, latter, gap)
        imgs, lab = synth_clip(11, frames, H, W, 3)
        rec = run_reference_clip(engine, imgs, lab, (H, W), capture_logits=(frames - 1,))
        meta = dict(H=H, W=W, frames=frames, gap=gap, former=former, latter=latter, seed=11,
                    indexes=rec["indexes"], ema=rec["ema"], visits=rec["visits"],
                    hist=rec["hist"], mass0=rec["mass0"],
                    label_sha=[sha(l) for l in rec["labels"]])
        with open(os.path.join(HERE, f"clip_small_{name}.json"), "w") as f:
            json.dump(meta, f)
        np.savez_compressed(os.path.join(HERE, f"clip_small_{name}.npz"),
                            labels=torch.stack(rec["labels"]).numpy(),
                            last_logits=rec["logits"][frames - 1].numpy())
        print("clip", name, "indexes", rec["indexes"][-1])

    # full 480p geometry (481x849 -> 31x54), K=4, gap=2 so an eviction occurs at frame 8
    cfg, model, engine = rh.build_reference("r50_deaotl", 1, 3, 2)
    H, W, frames = 481, 849, 10
    imgs, lab = synth_clip(0, frames, H, W, 3)
    rec = run_reference_clip(engine, imgs, lab, (480, 854), capture_logits=(1, 8, 9))
    meta = dict(H=H, W=W, out_hw=[480, 854], frames=frames, gap=2, former=1, latter=3, seed=0,
                indexes=rec["indexes"], ema=rec["ema"], visits=rec["visits"], hist=rec["hist"],
                mass0=rec["mass0"], label_sha=[sha(l) for l in rec["labels"]])
    with open(os.path.join(HERE, "clip_480p.json"), "w") as f:
        json.dump(meta, f)
    np.savez_compressed(os.path.join(HERE, "clip_480p.npz"),
                        **{f"logits_{t}": v.numpy().astype(np.float16) for t, v in rec["logits"].items()},
                        labels=torch.stack(rec["labels"]).numpy())
    print("clip 480p indexes", rec["indexes"])


def gen_ignore_clip():
    """A clip whose reference mask holds 255 pixels, through the reference's own
    add_reference_frame (no ignore mask there) and update_memory (argmax labels, no 255)."""
    name, H, W, frames, gap, former, latter = "ign255_k4_gap2", 97, 129, 10, 2, 1, 3
    cfg, model, engine = rh.build_reference("r50_deaotl", former, latter, gap)
    imgs, lab = synth_clip(11, frames, H, W, 3)
    lab = ignore_region_label(lab)
    rec = dict(indexes=[], labels=[], logits={})
    with torch.no_grad(), rh.quiet():
        engine.restart_engine()
        engine.add_reference_frame(imgs[0], lab.int(), obj_nums=[3], frame_step=0)
        sub = engine.aot_engines[0]
        rec["ref_logits"] = sub.pred_id_logits.clone()
        for t in range(1, frames):
            logit = engine.match_propogate_one_frame(imgs[t], output_size=(H, W))
            pred = torch.argmax(torch.softmax(logit, dim=1), dim=1, keepdim=True).float()
            engine.update_memory(F.interpolate(pred, size=engine.input_size_2d, mode="nearest"))
            rec["indexes"].append(list(sub.long_memories_indexes))
            rec["labels"].append(pred[0, 0].to(torch.uint8))
        rec["last_logits"] = sub.pred_id_logits.clone()
    meta = dict(H=H, W=W, frames=frames, gap=gap, former=former, latter=latter, seed=11,
                indexes=rec["indexes"], label_sha=[sha(l) for l in rec["labels"]])
    json.dump(meta, open(os.path.join(HERE, f"clip_small_{name}.json"), "w"))
    np.savez_compressed(os.path.join(HERE, f"clip_small_{name}.npz"), labels=torch.stack(rec["labels"]).numpy(),
                        ref_logits=rec["ref_logits"].numpy(), last_logits=rec["last_logits"].numpy())
    print("ignore clip indexes", rec["indexes"][-1])


def gen_tta():
    """Flip test-time augmentation + mid-clip new object, driven exactly as
    managers/evaluator.py:337-527 drives its engines (one engine per augmentation, the
    second on a deep copy of the model; probabilities un-flipped and averaged; labels
    flipped back per engine).  The new object arrives at frame 10 of 12 with gap=2 so that
    no long-term update follows it (the reference raises in restrict_long_memories after a
    mid-clip re-reference: transformer.py:954 size mismatch)."""
    import copy
    ref = rh.import_reference()
    H, W, frames, gap, out_hw, new_at = 97, 129, 12, 2, (90, 120), 10
    cfg, model, eng0 = rh.build_reference("r50_deaotl", 1, 3, gap)
    eng1 = ref["build_engine"](cfg.MODEL_ENGINE, phase="eval", aot_model=copy.deepcopy(model),
                               gpu_id=0, long_term_mem_gap=gap)
    eng1.eval()
    engines, flips = [eng0, eng1], [False, True]
    imgs, lab = synth_clip(23, frames, H, W, 3)
    lab0 = F.interpolate(lab.float(), size=out_hw, mode="nearest")      # dataset label at original size
    labels, indexes = [], []
    with torch.no_grad(), rh.quiet():
        for e in engines:
            e.restart_engine()
            e.long_term_mem_gap = gap
        for t in range(frames):
            all_preds, new_obj_label = [], None
            cur_label = lab0 if t == 0 else (tta_new_object_label(out_hw) if t == new_at else None)
            for e, fl in zip(engines, flips):
                img = imgs[t].flip(3) if fl else imgs[t]
                cl = None if cur_label is None else (cur_label.flip(3) if fl else cur_label)
                if t == 0:
                    _l = F.interpolate(cl, size=img.shape[2:], mode="nearest").int()
                    e.add_reference_frame(img, _l, frame_step=0, obj_nums=[3])
                else:
                    logit = e.match_propogate_one_frame(img, output_size=out_hw)
                    if fl:
                        logit = logit.flip(3)
                    all_preds.append(torch.softmax(logit, dim=1))
                    if not fl and cl is not None and new_obj_label is None:
                        new_obj_label = cl
            if t == 0:
                continue
            prob = torch.mean(torch.cat(all_preds, dim=0), dim=0, keepdim=True)
            pred = torch.argmax(prob, dim=1, keepdim=True).float()
            if new_obj_label is not None:
                keep = (new_obj_label == 0).float()
                pred = pred * keep + new_obj_label * (1 - keep)
                new_nums = [int(pred.max().item())]
                for e, fl in zip(engines, flips):
                    img = imgs[t].flip(3) if fl else imgs[t]
                    cl = F.interpolate(pred.flip(3) if fl else pred, size=e.input_size_2d, mode="nearest")
                    e.add_reference_frame(img, cl, obj_nums=new_nums, frame_step=t)
            else:
                for e, fl in zip(engines, flips):
                    e.update_memory(F.interpolate(pred.flip(3) if fl else pred, size=e.input_size_2d,
                                                  mode="nearest"))
            labels.append(pred[0, 0].to(torch.uint8))
            indexes.append([list(e.aot_engines[0].long_memories_indexes) for e in engines])
    meta = dict(H=H, W=W, out_hw=list(out_hw), frames=frames, gap=gap, former=1, latter=3, seed=23,
                flips=flips, new_at=new_at, indexes=indexes, label_sha=[sha(l) for l in labels],
                palette_sha=hashlib.sha256(bytes(__import__("importlib").import_module("utils.image")._palette)).hexdigest())
    with open(os.path.join(HERE, "clip_tta_k4_gap2.json"), "w") as f:
        json.dump(meta, f)
    np.savez_compressed(os.path.join(HERE, "clip_tta_k4_gap2.npz"), labels=torch.stack(labels).numpy())
    print("tta clip indexes", indexes[-1], "labels max", int(torch.stack(labels).max()))


def gen_tta_multiscale():
    """Multi-scale x flip test-time augmentation (cfg.TEST_MULTISCALE = [1.0, 1.3] with TEST_FLIP: four engines, in the
    order dataloaders/video_transforms.py:563-652 emits the samples -- per scale the plain copy, then the flipped one),
    driven as managers/evaluator.py:337-527 drives them: logits of every engine resized to the original size, un-flipped,
    softmaxed, averaged; the label flipped / nearest-resized back to every engine's own input size.  New object at
    frame 6 of 8, gap 2 (no long-term update may follow a mid-clip re-reference in the reference, see gen_tta)."""
    import copy
    ref = rh.import_reference()
    H, W, frames, gap, out_hw, new_at, scale = 97, 129, 8, 2, (90, 120), 6, 1.3
    from rmem_amd.driver import restrict_size
    H2, W2 = restrict_size(H, W, max_size=800, scale=scale, align_corners=True)
    cfg, model, eng0 = rh.build_reference("r50_deaotl", 1, 3, gap)
    engines = [eng0]
    for _ in range(3):
        e = ref["build_engine"](cfg.MODEL_ENGINE, phase="eval", aot_model=copy.deepcopy(model), gpu_id=0,
                                long_term_mem_gap=gap)
        e.eval()
        engines.append(e)
    flips = [False, True, False, True]
    imgs, lab = synth_clip(29, frames, H, W, 3)
    big = tta_scaled_images(imgs, (H2, W2))
    src = [imgs, imgs, big, big]
    lab0 = F.interpolate(lab.float(), size=out_hw, mode="nearest")
    labels, indexes = [], []
    with torch.no_grad(), rh.quiet():
        for e in engines:
            e.restart_engine()
            e.long_term_mem_gap = gap
        for t in range(frames):
            all_preds, new_obj_label = [], None
            cur_label = lab0 if t == 0 else (tta_new_object_label(out_hw) if t == new_at else None)
            for e, fl, sr in zip(engines, flips, src):
                img = sr[t].flip(3) if fl else sr[t]
                cl = None if cur_label is None else (cur_label.flip(3) if fl else cur_label)
                if t == 0:
                    _l = F.interpolate(cl, size=img.shape[2:], mode="nearest").int()
                    e.add_reference_frame(img, _l, frame_step=0, obj_nums=[3])
                else:
                    logit = e.match_propogate_one_frame(img, output_size=out_hw)
                    if fl:
                        logit = logit.flip(3)
                    all_preds.append(torch.softmax(logit, dim=1))
                    if not fl and cl is not None and new_obj_label is None:
                        new_obj_label = cl
            if t == 0:
                continue
            prob = torch.mean(torch.cat(all_preds, dim=0), dim=0, keepdim=True)
            pred = torch.argmax(prob, dim=1, keepdim=True).float()
            if new_obj_label is not None:
                keep = (new_obj_label == 0).float()
                pred = pred * keep + new_obj_label * (1 - keep)
                new_nums = [int(pred.max().item())]
                for e, fl, sr in zip(engines, flips, src):
                    img = sr[t].flip(3) if fl else sr[t]
                    cl = F.interpolate(pred.flip(3) if fl else pred, size=e.input_size_2d, mode="nearest")
                    e.add_reference_frame(img, cl, obj_nums=new_nums, frame_step=t)
            else:
                for e, fl in zip(engines, flips):
                    e.update_memory(F.interpolate(pred.flip(3) if fl else pred, size=e.input_size_2d, mode="nearest"))
            labels.append(pred[0, 0].to(torch.uint8))
            indexes.append([list(e.aot_engines[0].long_memories_indexes) for e in engines])
    meta = dict(H=H, W=W, H2=H2, W2=W2, scale=scale, out_hw=list(out_hw), frames=frames, gap=gap, former=1, latter=3,
                seed=29, flips=flips, new_at=new_at, indexes=indexes, label_sha=[sha(l) for l in labels],
                input_sizes=[list(e.input_size_2d) for e in engines])
    with open(os.path.join(HERE, "clip_tta_ms_gap2.json"), "w") as f:
        json.dump(meta, f)
    np.savez_compressed(os.path.join(HERE, "clip_tta_ms_gap2.npz"), labels=torch.stack(labels).numpy())
    print("multi-scale tta clip: sizes", meta["input_sizes"], "indexes", indexes[-1], "labels max", int(torch.stack(labels).max()))


def gen_multiengine():
    """the multi-engine wrapper's pure tensor functions, called on the imported reference
    itself (AOTInferEngine.separate_mask, aot_engine.py:604-618 label branch; soft_logit_aggregation,
    :650-673) with a dummy `aot_engines` list of length 2 / 3 -- the reference's shared-LSTT crash
    (see the note in this file) is never reached.  Inputs are seeded here and stored with the outputs."""
    cfg, model, engine = rh.build_reference("r50_deaotl", 1, 3, 5)
    g = torch.Generator().manual_seed(8604)
    out = {}
    for n_eng, hw in ((2, (37, 53)), (3, (21, 29))):
        engine.aot_engines = [object()] * n_eng           # only len() is read by the two functions
        # label maps with ids 0 .. n_eng*10 (+ beyond the last engine's range) and the ignore id 255
        ids = torch.randint(0, n_eng * engine.max_aot_obj_num + 3, (1, 1) + hw, generator=g).float()
        ids[0, 0, :3, :5] = 255.0
        sep = engine.separate_mask(ids)
        sep3 = engine.separate_mask(ids[0])               # 3-d mask branch (len(mask.size()) == 3)
        logits = [torch.randn(1, engine.max_aot_obj_num + 1, *hw, generator=g) * 6.0 for _ in range(n_eng)]
        logits[0][0, 0, :4] = 40.0                         # saturates the clamp(1e-5, 1 - 1e-5) on both sides
        logits[1][0, 3, 5:9] = 40.0
        agg = engine.soft_logit_aggregation(logits)
        k = f"e{n_eng}"
        out[k + "_mask"] = ids.numpy()
        for i, (a, b) in enumerate(zip(sep, sep3)):
            out[f"{k}_sep{i}"] = a.numpy()
            out[f"{k}_sep3d{i}"] = b.numpy()
        for i, lg in enumerate(logits):
            out[f"{k}_logit{i}"] = lg.numpy()
        out[k + "_agg"] = agg.numpy()
    # single-engine fast paths return their input (aot_engine.py:607-608, 651-652)
    engine.aot_engines = [object()]
    m1 = torch.randint(0, 11, (1, 1, 9, 11), generator=g).float()
    l1 = torch.randn(1, 11, 9, 11, generator=g)
    assert engine.separate_mask(m1)[0] is m1 and engine.soft_logit_aggregation([l1]) is l1
    engine.aot_engines = []
    np.savez_compressed(os.path.join(HERE, "multiengine_wrapper.npz"), **out)
    json.dump({"max_aot_obj_num": int(engine.max_aot_obj_num), "cases": {"e2": [37, 53], "e3": [21, 29]},
               "single_engine_identity": True, "seed": 8604},
              open(os.path.join(HERE, "multiengine_wrapper.json"), "w"))
    print("multi-engine wrapper vectors:", sorted(out)[:6], "...")


def gen_clip_480p_fp64(tie_margin=1e-4, token_stride=4):
    """fp64 arbitration of the golden 481x849 clip.  The reference itself is run
    with a double-precision model (``torch.set_default_dtype(float64)`` while it is built; the name-keyed fp32
    weights are cast exactly), teacher-forced with the labels its fp32 run produced (clip_480p.npz), and then
    once more in fp32 to record how far the fp32 CPU path itself is from fp64.  Stored (data only):
      labels64            label maps of the fp64 run
      tie_idx_t / tie_cls_t / tie_l64_t / tie_l32_t
                          every output pixel whose two best class logits are closer than ``tie_margin`` in fp64:
                          flat index, the two classes, their fp64 logits, the fp32 reference's logits
      lstt64_t            final LSTT output (after the GroupNorm, what the decoder reads) of the fp64 run at
                          tokens t % stride :: stride, rounded to fp32
      lstt32_err_t, dec32_err_t
                          [max, rms] of |fp32 reference - fp64| over the whole LSTT output / decoder logits
    """
    ref = rh.import_reference()
    meta = json.load(open(os.path.join(HERE, "clip_480p.json")))
    gold = np.load(os.path.join(HERE, "clip_480p.npz"))
    H, W, frames, out_hw = meta["H"], meta["W"], meta["frames"], tuple(meta["out_hw"])
    imgs32, lab = synth_clip(meta["seed"], frames, H, W, 3)
    AOTEngine = ref["aot_engine"].AOTEngine
    prev_assign = AOTEngine.assign_identity

    def run(dtype):
        # utils/image.py:74 hard-codes .float() for the one-hot planes: cast them (0/1, exact) to the model's dtype
        AOTEngine.assign_identity = lambda self, oh, ign=None: prev_assign(
            self, oh.to(dtype), None if ign is None else ign.to(dtype))
        torch.set_default_dtype(dtype)
        try:
            cfg, model, engine = rh.build_reference("r50_deaotl", meta["former"], meta["latter"], meta["gap"])
            assert next(model.parameters()).dtype == dtype
            got = {}
            hook = model.LSTT.register_forward_hook(lambda m, i, o: got.__setitem__("lstt", o[-1][:, 0].clone()))
            rec = dict(up=[], lstt=[], dec=[], idx=[])
            with torch.no_grad(), rh.quiet():
                engine.restart_engine()
                engine.add_reference_frame(imgs32[0].to(dtype), lab.int(), obj_nums=[int(lab.max())], frame_step=0)
                sub = engine.aot_engines[0]
                for t in range(1, frames):
                    up = engine.match_propogate_one_frame(imgs32[t].to(dtype), output_size=out_hw)
                    rec["up"].append(up[0].clone())
                    rec["lstt"].append(got["lstt"])
                    rec["dec"].append(sub.pred_id_logits.clone())
                    fed = torch.from_numpy(gold["labels"][t - 1]).to(dtype)[None, None]
                    engine.update_memory(F.interpolate(fed, size=engine.input_size_2d, mode="nearest"))
                    rec["idx"].append(list(sub.long_memories_indexes))
            hook.remove()
            return rec
        finally:
            torch.set_default_dtype(torch.float32)
            AOTEngine.assign_identity = prev_assign

    r64, r32 = run(torch.float64), run(torch.float32)
    assert r64["idx"] == meta["indexes"] and r32["idx"] == meta["indexes"]
    out, info = {}, dict(tie_margin=tie_margin, token_stride=token_stride, mism32_vs_64=[], n_tie=[],
                         lstt32_err=[], dec32_err=[])
    labels64 = []
    for t in range(1, frames):
        u64, u32 = r64["up"][t - 1], r32["up"][t - 1]
        l32 = torch.argmax(u32, dim=0).to(torch.uint8)
        assert torch.equal(l32, torch.from_numpy(gold["labels"][t - 1])), "the fp32 re-run must reproduce the golden labels"
        l64 = torch.argmax(u64, dim=0).to(torch.uint8)
        labels64.append(l64)
        top = torch.topk(u64, 2, dim=0)
        tie = torch.nonzero(((top.values[0] - top.values[1]) < tie_margin).flatten())[:, 0]
        cls = top.indices.flatten(1)[:, tie].T.contiguous()                   # [n, 2]
        f64 = u64.flatten(1)[:, tie]
        f32 = u32.flatten(1)[:, tie]
        ar = torch.arange(tie.numel())
        out[f"tie_idx_{t}"] = tie.to(torch.int32).numpy()
        out[f"tie_cls_{t}"] = cls.to(torch.uint8).numpy()
        out[f"tie_l64_{t}"] = torch.stack([f64[cls[:, 0], ar], f64[cls[:, 1], ar]], 1).numpy()
        out[f"tie_l32_{t}"] = torch.stack([f32[cls[:, 0], ar], f32[cls[:, 1], ar]], 1).numpy()
        out[f"lstt64_{t}"] = r64["lstt"][t - 1][t % token_stride::token_stride].float().numpy()
        e = (r32["lstt"][t - 1].double() - r64["lstt"][t - 1]).abs()
        d = (r32["dec"][t - 1].double() - r64["dec"][t - 1])[:, :int(lab.max()) + 1].abs()
        out[f"lstt32_err_{t}"] = np.array([float(e.max()), float((e ** 2).mean().sqrt())])
        out[f"dec32_err_{t}"] = np.array([float(d.max()), float((d ** 2).mean().sqrt())])
        mm = torch.nonzero((l32 != l64).flatten())[:, 0]
        assert all(int(i) in set(tie.tolist()) for i in mm), "an fp32/fp64 disagreement outside the near-tie set"
        info["mism32_vs_64"].append(int(mm.numel()))
        info["n_tie"].append(int(tie.numel()))
        info["lstt32_err"].append(out[f"lstt32_err_{t}"].tolist())
        info["dec32_err"].append(out[f"dec32_err_{t}"].tolist())
    out["labels64"] = torch.stack(labels64).numpy()
    np.savez_compressed(os.path.join(HERE, "clip_480p_fp64.npz"), **out)
    json.dump(info, open(os.path.join(HERE, "clip_480p_fp64.json"), "w"))
    print("480p fp64 arbitration:", info)


def gen_clip_480p_long(frames=46, tie_margin=1e-4, resume=False):
    """BASELINE.json configs[1] at the evaluator's own schedule (managers/evaluator.py:331-332: gap 5), K = 4,
    481x849, long enough for the benchmarked steady state: the bank fills at frame 15 and every long-term update
    from frame 20 on evicts (frames 20, 25, 30, 35, 40, 45 -> six evictions in 46 frames).
      clip_480p_long.json/.npz   the reference's closed-loop fp32 run: label maps, long_memories_indexes /
                                 EMA / visit dictionaries / layer-0 attention mass after every frame, decoder
                                 logits (fp16) of frames 1, 20 (first eviction) and the last frame
      clip_480p_long_fp64.npz    the same clip through the reference in DOUBLE precision, teacher-forced with those
                                 labels: per frame the near-tie list (flat pixel index, the two best classes, their
                                 fp64 logits and the fp32 re-run's logits) and the pixels on which the fp32
                                 reference's own label differs from the fp64 one (always inside the tie list)
    """
    _gen_long_clip("480p_long", 481, 849, (480, 854), 5, 1, 3, 0, frames, (1, 20, frames - 1), tie_margin, resume)


def gen_clip_720p_k8(frames=11, tie_margin=1e-4, resume=False):
    """BASELINE.json configs[2] through the reference itself: 721x1281 (46x81 = 3726 tokens), K = 8 (former 1 + latter 7),
    3 objects, gap 1 so that the bank passes four slots (temporal-PE rows for T > 4), fills at T = 8 and evicts inside 11
    frames -- the schedule tests/test_hip_engine.py::test_720p_k8_vs_oracle runs against the oracle.  Same files as the long
    480p clip: clip_720p_k8.json/.npz (closed-loop fp32 run) and clip_720p_k8_fp64.* (double-precision near-tie lists)."""
    _gen_long_clip("720p_k8", 721, 1281, (720, 1280), 1, 1, 7, 11, frames, (1, frames - 1), tie_margin, resume)


def _gen_long_clip(tag, H, W, out_hw, gap, former, latter, seed, frames, cap, tie_margin=1e-4, resume=False,
                   model_name="r50_deaotl"):
    """A closed-loop run of the reference at one of BASELINE.json's full sizes plus its double-precision arbitration:
      clip_{tag}.json/.npz       the reference's closed-loop fp32 run: label maps, long_memories_indexes / EMA / visit
                                 dictionaries / layer-0 attention mass after every frame, decoder logits (fp16) of the
                                 frames in `cap`
      clip_{tag}_fp64.npz        the same clip through the reference in DOUBLE precision, teacher-forced with those
                                 labels: per frame the near-tie list (flat pixel index, the two best classes, their
                                 fp64 logits and the fp32 re-run's logits) and the pixels on which the fp32
                                 reference's own label differs from the fp64 one (always inside the tie list)
    """
    ref = rh.import_reference()
    imgs32, lab = synth_clip(seed, frames, H, W, 3)
    if resume:             # (--long-fp64-only: the closed-loop run is on disk, redo the arbitration only)
        meta = json.load(open(os.path.join(HERE, f"clip_{tag}.json")))
        gold_labels = np.load(os.path.join(HERE, f"clip_{tag}.npz"))["labels"]
        assert meta["frames"] == frames
    else:
        cfg, model, engine = rh.build_reference(model_name, former, latter, gap)
        rec = run_reference_clip(engine, imgs32, lab, out_hw, capture_logits=cap)
        n_evict = sum(1 for a, b in zip(rec["indexes"][:-1], rec["indexes"][1:]) if a != b and len(b) <= len(a))
        meta = dict(H=H, W=W, out_hw=list(out_hw), frames=frames, gap=gap, former=former, latter=latter, seed=seed,
                    indexes=rec["indexes"], ema=rec["ema"], visits=rec["visits"], hist=rec["hist"], mass0=rec["mass0"],
                    label_sha=[sha(l) for l in rec["labels"]], evictions=n_evict, logit_frames=list(cap))
        json.dump(meta, open(os.path.join(HERE, f"clip_{tag}.json"), "w"))
        gold_labels = torch.stack(rec["labels"]).numpy()
        np.savez_compressed(os.path.join(HERE, f"clip_{tag}.npz"),
                            **{f"logits_{t}": v.numpy().astype(np.float16) for t, v in rec["logits"].items()},
                            labels=gold_labels)
        print(f"clip {tag}: evictions", n_evict, "indexes", rec["indexes"][-1], flush=True)

    AOTEngine = ref["aot_engine"].AOTEngine
    prev_assign = AOTEngine.assign_identity

    def run(dtype):
        AOTEngine.assign_identity = lambda self, oh, ign=None: prev_assign(
            self, oh.to(dtype), None if ign is None else ign.to(dtype))
        torch.set_default_dtype(dtype)
        try:
            cfg, model, engine = rh.build_reference(model_name, former, latter, gap)
            ups, idx = [], []
            with torch.no_grad(), rh.quiet():
                engine.restart_engine()
                engine.add_reference_frame(imgs32[0].to(dtype), lab.int(), obj_nums=[int(lab.max())], frame_step=0)
                sub = engine.aot_engines[0]
                for t in range(1, frames):
                    up = engine.match_propogate_one_frame(imgs32[t].to(dtype), output_size=out_hw)
                    ups.append(up[0].clone())
                    fed = torch.from_numpy(gold_labels[t - 1]).to(dtype)[None, None]
                    engine.update_memory(F.interpolate(fed, size=engine.input_size_2d, mode="nearest"))
                    idx.append(list(sub.long_memories_indexes))
            return ups, idx
        finally:
            torch.set_default_dtype(torch.float32)
            AOTEngine.assign_identity = prev_assign

    u64s, idx64 = run(torch.float64)
    assert idx64 == meta["indexes"], "the fp64 run must evict the same slots"
    # (the fp32 side is the closed-loop run above: its own labels are what it was fed, so a teacher-forced fp32 re-run
    # reproduces it frame by frame; its full-size logits are recomputed here only for the tie pixels)
    u32s, idx32 = run(torch.float32)
    assert idx32 == meta["indexes"]
    out, info = {}, dict(tie_margin=tie_margin, mism32_vs_64=[], n_tie=[])
    for t in range(1, frames):
        u64, u32 = u64s[t - 1], u32s[t - 1]
        # the label the evaluator takes is argmax(softmax(logit)) (managers/evaluator.py:430-441): in fp32 the softmax can
        # round two near-tied logits to one probability, and argmax then takes the lower id
        l32 = torch.argmax(torch.softmax(u32, dim=0), dim=0).to(torch.uint8)
        n_bad = int((l32 != torch.from_numpy(gold_labels[t - 1])).sum())
        assert n_bad == 0, f"frame {t}: the fp32 re-run must reproduce the golden labels ({n_bad} pixels differ)"
        l64 = torch.argmax(u64, dim=0).to(torch.uint8)
        top = torch.topk(u64, 2, dim=0)
        tie = torch.nonzero(((top.values[0] - top.values[1]) < tie_margin).flatten())[:, 0]
        cls = top.indices.flatten(1)[:, tie].T.contiguous()
        f64 = u64.flatten(1)[:, tie]
        f32 = u32.flatten(1)[:, tie]
        ar = torch.arange(tie.numel())
        out[f"tie_idx_{t}"] = tie.to(torch.int32).numpy()
        out[f"tie_cls_{t}"] = cls.to(torch.uint8).numpy()
        out[f"tie_l64_{t}"] = torch.stack([f64[cls[:, 0], ar], f64[cls[:, 1], ar]], 1).numpy()
        out[f"tie_l32_{t}"] = torch.stack([f32[cls[:, 0], ar], f32[cls[:, 1], ar]], 1).numpy()
        mm = torch.nonzero((l32 != l64).flatten())[:, 0]
        assert all(int(i) in set(tie.tolist()) for i in mm), "an fp32/fp64 disagreement outside the near-tie set"
        out[f"mism32_idx_{t}"] = mm.to(torch.int32).numpy()
        out[f"mism32_l64_{t}"] = l64.flatten()[mm].numpy()
        info["mism32_vs_64"].append(int(mm.numel()))
        info["n_tie"].append(int(tie.numel()))
    np.savez_compressed(os.path.join(HERE, f"clip_{tag}_fp64.npz"), **out)
    json.dump(info, open(os.path.join(HERE, f"clip_{tag}_fp64.json"), "w"))
    print(f"{tag} fp64 arbitration:", info, flush=True)


def gen_aot_fp64_lists():
    """Double-precision near-tie lists for the AOT / Swin clips that already hold the reference's fp32 run (clip_aot_480p:
    BASELINE.json configs[0], 481x849 x 16 frames; clip_swin_k4_gap2: the small Swin clip), so that their parity tests can
    assert the near-tie property instead of pixel budgets -- and the reference's own run of configs[4] at its full size
    (SwinB-AOTL + RMem, 480x848 -> 30x53 tokens, gap 1, 10 frames: the bank fills at frame 4, every later frame evicts)."""
    part = os.environ.get("GOLDEN_PART", "all")      # the fp32 re-run of a stored clip must reproduce its labels bit for bit, which
    # needs the thread count of the run that stored it (CPU GEMM reduction order): OMP_NUM_THREADS=7 for aot_480p,
    # 8 for swin_k4_gap2, 7 for swin_480p (GOLDEN_RESUME=1 redoes only the fp64 part of swin_480p)
    if part in ("all", "aot"):
        _gen_long_clip("aot_480p", 481, 849, (480, 854), 5, 1, 3, 0, 16, (1, 15), resume=True, model_name="r50_aotl")
    if part in ("all", "small"):        # the small DeAOT / AOT clips: their fp32 runs are on disk (gen_clips, gen_aot_clips)
        for name, H, W, frames, gap, former, latter in (("k4_gap2", 97, 129, 16, 2, 1, 3), ("k4_gap5", 97, 129, 24, 5, 1, 3),
                                                        ("k8_gap2", 81, 113, 28, 2, 1, 7), ("k2_gap1", 81, 97, 10, 1, 1, 1)):
            _gen_long_clip("small_" + name, H, W, (H, W), gap, former, latter, 11, frames, (frames - 1,), resume=True)
        for name, H, W, frames, gap, former, latter in (("aot_k4_gap2", 97, 129, 16, 2, 1, 3), ("aot_k2_gap1", 81, 97, 10, 1, 1, 1)):
            _gen_long_clip(name, H, W, (H, W), gap, former, latter, 11, frames, (frames - 1,), resume=True, model_name="r50_aotl")
    if part in ("all", "swin_small"):
        _gen_long_clip("swin_k4_gap2", 128, 160, (128, 160), 2, 1, 3, 7, 12, (11,), resume=True, model_name="swinb_aotl")
    if part in ("all", "swin_480p"):
        # (tie list to 1e-3: on this backbone the fp32 reference itself leaves its fp64 run at margins up to 1.4e-4)
        _gen_long_clip("swin_480p", 480, 848, (480, 854), 1, 1, 3, 5, 10, (1, 9), tie_margin=1e-3, model_name="swinb_aotl",
                       resume=os.environ.get("GOLDEN_RESUME") == "1")



def gen_load_network_cases():
    """The reference's OWN load_network (utils/checkpoint.py:75-101) run on CPU over the payload variants of
    tests/golden/inputs.py:ckpt_cases -- torch.load's hard-coded CUDA map_location and net.cuda(gpu) are redirected from
    outside the tree (the file still goes through torch.save / torch.load).  Stored per case: every tensor of the
    resulting state_dict and the returned list of removed keys."""
    import tempfile
    from inputs import ckpt_cases, ckpt_toy_net
    rh.import_reference()
    from utils.checkpoint import load_network as ref_load_network
    orig_load, orig_cuda = torch.load, torch.nn.Module.cuda
    torch.load = lambda f, map_location=None, **kw: orig_load(f, map_location="cpu", **kw)
    torch.nn.Module.cuda = lambda self, device=None: self
    out, meta = {}, {}
    try:
        with tempfile.TemporaryDirectory() as d:
            for name, ckpt in ckpt_cases().items():
                path = os.path.join(d, name + ".pth")
                torch.save(ckpt, path)
                net = ckpt_toy_net(0)
                net, removed = ref_load_network(net, path, 0)
                for k, v in net.state_dict().items():
                    out[f"{name}/{k}"] = v.detach().numpy()
                meta[name] = list(removed)
    finally:
        torch.load, torch.nn.Module.cuda = orig_load, orig_cuda
    np.savez_compressed(os.path.join(HERE, "load_network_cases.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "load_network_cases.json"), "w"), indent=0, sort_keys=True)
    print("load_network cases:", {k: v for k, v in meta.items()})


def gen_amp_clips():
    """The reference's reduced-precision mode (`--amp`: tools/eval.py:45-47,90-92 wraps the whole evaluation in
    torch.cuda.amp.autocast) on the golden clips, TEACHER-FORCED with the labels of its fp32 run (so that every frame
    is a statement about one forward pass, not about the closed loop's chaos).

    The container has no GPU, so the run is `torch.autocast("cpu", dtype=torch.float16)`.  CPU autocast does not
    promote softmax to fp32 as CUDA autocast does, and the reference needs that promotion: `local2global`
    (attention.py:390-396) index-puts the softmax output into an fp32 buffer and raises on CPU otherwise.  The
    harness therefore wraps `torch.softmax` for the duration of the run to take its input as fp32 -- CUDA autocast's
    rule for this op; nothing else is touched.  Stored: the autocast run's label maps, its mismatch count against
    the fp32 golden maps per frame, the decoder logits of the last frame (fp16)."""
    cases = [("clip_small_k4_gap2", None), ("clip_480p", None)]
    orig_softmax = torch.softmax
    torch.softmax = lambda x, dim, **kw: orig_softmax(x.float(), dim, **kw)
    try:
        for name, _ in cases:
            meta = json.load(open(os.path.join(HERE, name + ".json")))
            gold = np.load(os.path.join(HERE, name + ".npz"))
            H, W, frames = meta["H"], meta["W"], meta["frames"]
            out_hw = tuple(meta.get("out_hw", (H, W)))
            imgs, lab = synth_clip(meta["seed"], frames, H, W, 3)
            cfg, model, engine = rh.build_reference("r50_deaotl", meta["former"], meta["latter"], meta["gap"])
            labels, mism, idx = [], [], []
            with torch.no_grad(), rh.quiet(), torch.autocast("cpu", dtype=torch.float16):
                engine.restart_engine()
                engine.add_reference_frame(imgs[0], lab.int(), obj_nums=[int(lab.max())], frame_step=0)
                sub = engine.aot_engines[0]
                for t in range(1, frames):
                    logit = engine.match_propogate_one_frame(imgs[t], output_size=out_hw)
                    pred = torch.argmax(torch.softmax(logit, dim=1), dim=1)[0].to(torch.uint8)
                    labels.append(pred)
                    mism.append(int((pred.numpy() != gold["labels"][t - 1]).sum()))
                    fed = torch.from_numpy(gold["labels"][t - 1]).float()[None, None]
                    engine.update_memory(F.interpolate(fed, size=engine.input_size_2d, mode="nearest"))
                    idx.append(list(sub.long_memories_indexes))
                last = sub.pred_id_logits.float().clone()
            np.savez_compressed(os.path.join(HERE, name + "_amp.npz"), labels_amp=torch.stack(labels).numpy(),
                                last_logits_amp=last.numpy().astype(np.float16))
            json.dump(dict(autocast="cpu/float16 + softmax taken in fp32 (CUDA autocast's rule)", mism_amp_vs_fp32=mism,
                           indexes=idx, indexes_equal_fp32=bool(idx == meta["indexes"])),
                      open(os.path.join(HERE, name + "_amp.json"), "w"))
            print(name, "autocast fp16 vs fp32 golden, mismatching pixels per frame:", mism, "indexes equal:", idx == meta["indexes"])
    finally:
        torch.softmax = orig_softmax


def main():
    torch.manual_seed(0)
    if "--tta-only" in sys.argv:
        gen_tta()
        return
    if "--tta-ms-only" in sys.argv:
        gen_tta_multiscale()
        return
    if "--ignore-only" in sys.argv:
        gen_ignore_clip()
        return
    if "--multiengine-only" in sys.argv:
        gen_multiengine()
        return
    if "--fp64-only" in sys.argv:
        gen_clip_480p_fp64()
        return
    if "--ckpt-only" in sys.argv:
        gen_load_network_cases()
        return
    if "--block-full-only" in sys.argv:
        cfg, model, engine = rh.build_reference("r50_deaotl", 1, 3, 5)
        gen_block_fullsize(model)
        return
    if "--blocks-only" in sys.argv:
        cfg, model, engine = rh.build_reference("r50_deaotl", 1, 3, 5)
        gen_blocks(model)
        return
    if "--long-only" in sys.argv or "--long-fp64-only" in sys.argv:
        gen_clip_480p_long(resume="--long-fp64-only" in sys.argv)
        return
    if "--aot-fp64-only" in sys.argv:
        gen_aot_fp64_lists()
        return
    if "--720p-only" in sys.argv or "--720p-fp64-only" in sys.argv:
        gen_clip_720p_k8(resume="--720p-fp64-only" in sys.argv)
        return
    if "--amp-only" in sys.argv:
        gen_amp_clips()
        return
    if "--aot-only" not in sys.argv and "--swin-only" not in sys.argv:
        cfg, model, engine = rh.build_reference("r50_deaotl", 1, 3, 5)
        gen_manifest(model)
        gen_blocks(model)
        gen_idassign(model)
        gen_clips()
    cfg, model, engine = rh.build_reference("r50_aotl", 1, 3, 5)
    man = {k: list(v.shape) for k, v in model.state_dict().items()}
    json.dump(man, open(os.path.join(HERE, "manifest_r50_aotl.json"), "w"), indent=0, sort_keys=True)
    if "--swin-only" not in sys.argv:
        gen_aot_blocks(model)
        gen_aot_clips()
    gen_swin()
    gen_tta()
    gen_tta_multiscale()
    gen_ignore_clip()
    gen_multiengine()
    gen_clip_480p_fp64()
    gen_clip_480p_long()
    gen_clip_720p_k8()
    gen_aot_fp64_lists()
    gen_load_network_cases()
    gen_amp_clips()
    os.system(f"du -sh {HERE}")


if __name__ == "__main__":
    main()
