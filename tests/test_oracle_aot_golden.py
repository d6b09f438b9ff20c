"""CPU: the AOT oracle (oracle/aot_ref.py) against the reference's golden vectors
(BASELINE.json configs[0])."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from inputs import AOT_BLOCK_CASES, aot_block_case_name, aot_block_inputs
from oracle import aot_ref as A
from oracle.engine_ref import OracleAOTEngine
from rmem_amd.config import get_config
from rmem_amd.model import build_vos_model
from rmem_amd.synth import load_synthetic_weights, synth_clip

TOL = 3e-5


@pytest.fixture(scope="module")
def aot_model():
    torch.manual_seed(0)
    m = build_vos_model("aot", get_config("r50_aotl")).eval()
    load_synthetic_weights(m)
    return m


def test_aot_manifest(aot_model, golden_dir):
    man = json.load(open(os.path.join(golden_dir, "manifest_r50_aotl.json")))
    assert {k: list(v.shape) for k, v in aot_model.state_dict().items()} == man


@pytest.mark.parametrize("case", AOT_BLOCK_CASES, ids=lambda c: aot_block_case_name(*c))
def test_aot_block_vs_golden(case, aot_model, golden_dir):
    layer, T, h, w, ref_frame = case
    gold = np.load(os.path.join(golden_dir, aot_block_case_name(*case) + ".npz"))
    i = aot_block_inputs(*case)
    sd = {k: v.detach() for k, v in aot_model.state_dict().items()}
    pos = A.sine_pos_emb(h, w)
    assert np.abs(pos.numpy() - gold["pos"]).max() < 1e-6
    mem = A.AOTMemory()
    if not ref_frame:
        mem.K, mem.V = list(i["bank_K"]), list(i["bank_V"])
        mem.sK, mem.sV = i["short_K"], i["short_V"]
    tgt, curr, mass, bank, short = A.aot_block(sd, layer, i["tgt"], mem, h, w, pos, sd["cur_pos_emb"][0],
                                               sd["mem_pos_emb"], i["id_emb"] if ref_frame else None)
    assert np.abs(tgt.numpy() - gold["out_tgt"]).max() < TOL
    assert np.abs(curr[0].numpy() - gold["curr_K"]).max() < TOL
    assert np.abs(curr[1].numpy() - gold["curr_V"]).max() < TOL
    assert np.abs(short[0].numpy() - gold["local_K"]).max() < TOL
    assert np.abs(short[1].numpy() - gold["local_V"]).max() < TOL
    if ref_frame:
        assert np.abs(bank[1][0].numpy() - gold["glob_V"]).max() < TOL
    else:
        assert np.abs(mass.numpy() - gold["mass"]).max() < TOL


def _run(meta, model, teacher=None):
    model.cfg = get_config("r50_aotl", meta["former"], meta["latter"])
    eng = OracleAOTEngine(model, long_term_mem_gap=meta["gap"])
    imgs, lab = synth_clip(meta["seed"], meta["frames"], meta["H"], meta["W"], 3)
    out_hw = tuple(meta.get("out_hw", (meta["H"], meta["W"])))
    eng.add_reference_frame(imgs[0], lab, obj_nums=[3], frame_step=0)
    rec = dict(indexes=[], labels=[], logits={})
    for t in range(1, meta["frames"]):
        logit = eng.match_propogate_one_frame(imgs[t], output_size=out_hw)
        pred = torch.argmax(torch.softmax(logit, dim=1), dim=1, keepdim=True).float()
        fed = pred if teacher is None else torch.from_numpy(teacher[t - 1]).float()[None, None]
        eng.update_memory(F.interpolate(fed, size=eng.input_size_2d, mode="nearest"))
        rec["indexes"].append(list(eng.long_memories_indexes))
        rec["labels"].append(pred[0, 0].to(torch.uint8))
        rec["logits"][t] = eng.pred_id_logits.clone()
    return rec


def _ties(golden_dir, fname):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from ties import Fp64Ties
    return Fp64Ties(np.load(os.path.join(golden_dir, fname)))


@pytest.mark.parametrize("name", ["aot_k4_gap2", "aot_k2_gap1"])
def test_aot_small_clip(name, aot_model, golden_dir):
    meta = json.load(open(os.path.join(golden_dir, f"clip_{name}.json")))
    gold = np.load(os.path.join(golden_dir, f"clip_{name}.npz"))
    rec = _run(meta, aot_model, teacher=gold["labels"])
    assert rec["indexes"] == meta["indexes"]
    labels = torch.stack(rec["labels"]).numpy()
    # every pixel off the reference's map: a near-tie of the reference's double-precision run that got one of the tie's two
    # classes (clip_*_fp64.npz, tests/ties.py) -- the property, not a pixel budget
    ties = _ties(golden_dir, f"clip_{name}_fp64.npz")
    mism = [ties.check(t + 1, labels[t], gold["labels"][t], 1e-5)[0] for t in range(labels.shape[0])]
    print(name, "pixels off the reference's maps per frame:", mism)
    assert np.abs(rec["logits"][meta["frames"] - 1].numpy() - gold["last_logits"]).max() < 1e-4


@pytest.mark.slow
def test_aot_480p_clip(aot_model, golden_dir):
    """BASELINE.json configs[0]: R50-AOTL + RMem, 481x849, 16 frames, K=4 (teacher-forced)."""
    meta = json.load(open(os.path.join(golden_dir, "clip_aot_480p.json")))
    gold = np.load(os.path.join(golden_dir, "clip_aot_480p.npz"))
    rec = _run(meta, aot_model, teacher=gold["labels"])
    assert rec["indexes"] == meta["indexes"]
    labels = torch.stack(rec["labels"]).numpy()
    # every pixel off the reference's map must be a near-tie of the reference's own double-precision run that received
    # one of the tie's two classes (clip_aot_480p_fp64.npz, tests/ties.py) -- the property, not a pixel budget
    ties = _ties(golden_dir, "clip_aot_480p_fp64.npz")
    mism = [ties.check(t + 1, labels[t], gold["labels"][t], 1e-5)[0] for t in range(labels.shape[0])]
    print("AOT 480p pixels off the reference's maps per frame, each an fp64 near-tie:", mism)
    for t in (1, 15):
        assert np.abs(rec["logits"][t].numpy() - gold[f"logits_{t}"].astype(np.float32)).max() < 2e-2


def test_swin_aot_clip(golden_dir):
    """BASELINE.json configs[4]: SwinB-AOTL + RMem.  Our Swin-B module (same state_dict keys)
    + the AOT oracle against the reference's encoder features and label maps (teacher-forced)."""
    meta = json.load(open(os.path.join(golden_dir, "clip_swin_k4_gap2.json")))
    gold = np.load(os.path.join(golden_dir, "clip_swin_k4_gap2.npz"))
    torch.manual_seed(0)
    model = build_vos_model("aot", get_config("swinb_aotl", meta["former"], meta["latter"])).eval()
    load_synthetic_weights(model)
    man = json.load(open(os.path.join(golden_dir, "manifest_swinb_aotl.json")))
    assert {k: list(v.shape) for k, v in model.state_dict().items()} == man
    imgs, lab = synth_clip(meta["seed"], meta["frames"], meta["H"], meta["W"], 3)
    with torch.no_grad():
        feats = model.encoder(imgs[0])
    assert np.abs(feats[2].numpy() - gold["enc16"].astype(np.float32)).max() < 5e-3      # fp16-stored gold
    assert np.abs(feats[0].mean(dim=(2, 3)).numpy() - gold["enc4_mean"]).max() < 1e-5
    eng = OracleAOTEngine(model, long_term_mem_gap=meta["gap"])
    eng.add_reference_frame(imgs[0], lab, obj_nums=[3], frame_step=0)
    ties = _ties(golden_dir, "clip_swin_k4_gap2_fp64.npz")
    mism, idx = [], []
    for t in range(1, meta["frames"]):
        logit = eng.match_propogate_one_frame(imgs[t], output_size=(meta["H"], meta["W"]))
        pred = torch.argmax(torch.softmax(logit, dim=1), dim=1)[0]
        mism.append(ties.check(t, pred.numpy().astype(np.uint8), gold["labels"][t - 1], 1e-5)[0])
        fed = torch.from_numpy(gold["labels"][t - 1]).float()[None, None]
        eng.update_memory(F.interpolate(fed, size=eng.input_size_2d, mode="nearest"))
        idx.append(list(eng.long_memories_indexes))
    assert idx == meta["indexes"]
    assert np.abs(eng.pred_id_logits.numpy() - gold["last_logits"]).max() < 1e-4
