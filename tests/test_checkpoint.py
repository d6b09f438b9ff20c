"""CPU: load_network remap rules (utils/checkpoint.py:75-101 of the reference)."""
import torch

from rmem_amd.checkpoint import load_network, remap_state_dict
from rmem_amd.config import get_config
from rmem_amd.model import build_vos_model
from rmem_amd.synth import load_synthetic_weights


def test_remap_rules():
    torch.manual_seed(0)
    src = build_vos_model("deaot", get_config("r50_deaotl")).eval()
    load_synthetic_weights(src)
    sd = src.state_dict()
    ckpt = {}
    for k, v in sd.items():
        ckpt["module." + k] = v.clone()                       # DataParallel prefix (rule 4)
    ckpt["module.patch_wise_id_bank.weight"] = sd["patch_wise_id_bank.weight"][:, :11].clone()   # rule 2
    ckpt["module.decoder.conv_out.weight"] = torch.zeros(7, 128, 1, 1)                           # shape mismatch
    ckpt["optimizer_junk"] = torch.zeros(3)
    dst = build_vos_model("deaot", get_config("r50_deaotl")).eval()
    before_last = dst.state_dict()["patch_wise_id_bank.weight"][:, 11].clone()
    before_out = dst.state_dict()["decoder.conv_out.weight"].clone()
    dst, removed = load_network(dst, {"state_dict": ckpt})
    new = dst.state_dict()
    for k, v in sd.items():
        if k in ("patch_wise_id_bank.weight", "decoder.conv_out.weight"):
            continue
        assert torch.equal(new[k], v), k
    # rule 2 is checked before the 'module.' retry in the reference, so a prefixed 11-channel bank is *not*
    # zero-extended: like every prefixed key whose stripped name does not match in shape it is dropped WITHOUT a
    # report (utils/checkpoint.py:93-95; pinned by tests/golden/load_network_cases.json: 'bank11_prefixed' -> [])
    assert torch.equal(new["decoder.conv_out.weight"], before_out)
    assert removed == ["optimizer_junk"]
    dropped = []
    remap_state_dict(dst.state_dict(), {"state_dict": ckpt}, dropped)
    assert set(dropped) == {"module.patch_wise_id_bank.weight", "module.decoder.conv_out.weight"}
    # un-prefixed 11-channel bank: first 11 input channels loaded, the 12th keeps the model's value
    out, removed2 = remap_state_dict(dst.state_dict(), {"model": {"patch_wise_id_bank.weight":
                                                                  sd["patch_wise_id_bank.weight"][:, :11] * 2}})
    assert removed2 == []
    assert torch.equal(out["patch_wise_id_bank.weight"][:, :11], sd["patch_wise_id_bank.weight"][:, :11] * 2)
    assert torch.equal(out["patch_wise_id_bank.weight"][:, 11], before_last)


def test_load_network_bumps_the_weights_version():
    """Engines pack the LSTT weights when they are built; load_network() afterwards must make them
    re-pack (rmem_amd/engine.py: update_size / restart_engine compare `_weights_version`)."""
    from rmem_amd.checkpoint import load_network
    from rmem_amd.config import get_config
    from rmem_amd.model import build_vos_model
    m = build_vos_model("deaot", get_config("r50_deaotl")).eval()
    v0 = m.__dict__.get("_weights_version", 0)
    m, _ = load_network(m, {k: v.clone() for k, v in m.state_dict().items()})
    assert m.__dict__["_weights_version"] == v0 + 1
    assert "_weights_version" not in m.state_dict()


def test_folding_is_idempotent_until_the_weights_change():
    """Every engine calls optimize_for_inference() when it is built.  An engine built earlier holds hipGraphs that
    replay the folded tensors, so a second call must NOT fold again (it used to: the second driver on one model freed
    what the first one's encoder graph read).  load_network() -- a new weights version -- and force=True fold again."""
    m = build_vos_model("deaot", get_config("r50_deaotl")).eval()
    load_synthetic_weights(m)
    m.optimize_for_inference(True)
    first = m.__dict__["_enc_infer"]
    ptr = first.conv1.weight.data_ptr() if hasattr(first, "conv1") else next(first.parameters()).data_ptr()
    m.optimize_for_inference(True)
    assert m.__dict__["_enc_infer"] is first
    again = m.__dict__["_enc_infer"]
    assert (again.conv1.weight.data_ptr() if hasattr(again, "conv1") else next(again.parameters()).data_ptr()) == ptr
    v0 = m.__dict__.get("_weights_version", 0)
    sd = {k: v.clone() * (0.5 if k.endswith("layer1.0.conv1.weight") else 1) for k, v in m.state_dict().items()}
    m, _ = load_network(m, sd)
    assert m.__dict__["_enc_infer"] is not first and m.__dict__["_weights_version"] == v0 + 1
    x = torch.randn(1, 3, 33, 49)
    with torch.no_grad():
        a, b = m.__dict__["_enc_infer"](x), m.encoder(x)
    assert all(torch.allclose(p, q, atol=1e-4, rtol=1e-4) for p, q in zip(a, b))      # folded from the NEW weights
    second = m.__dict__["_enc_infer"]
    # weights written in place WITHOUT load_network (load_state_dict by hand): the next call notices and folds again
    sd2 = {k: v.clone() * (2.0 if k.endswith("layer1.0.conv1.weight") else 1) for k, v in m.state_dict().items()}
    m.load_state_dict(sd2)
    v1 = m.__dict__["_weights_version"]
    m.optimize_for_inference(True)
    assert m.__dict__["_enc_infer"] is not second and m.__dict__["_weights_version"] == v1 + 1
    with torch.no_grad():
        a, b = m.__dict__["_enc_infer"](x), m.encoder(x)
    assert all(torch.allclose(p, q, atol=1e-4, rtol=1e-4) for p, q in zip(a, b))
    second = m.__dict__["_enc_infer"]
    v0 = v1
    m.optimize_for_inference(True, force=True)
    assert m.__dict__["_enc_infer"] is not second and m.__dict__["_weights_version"] == v0 + 2
    m.optimize_for_inference(False)
    assert m.__dict__["_enc_infer"] is None
    assert "_enc_infer_state" not in m.state_dict()


def test_remap_equals_the_reference_on_every_payload_variant(golden_dir):
    """tests/golden/load_network_cases.* = the reference's OWN load_network (utils/checkpoint.py:75-101) run on CPU over
    ten payload variants (make_golden.py:gen_load_network_cases): plain, 'state_dict' / 'model' wrappers (and both),
    'module.' prefixes, an 11-channel id bank plain and prefixed, shape mismatches plain and prefixed, mixed / doubly
    prefixed keys.  remap_state_dict must produce IDENTICAL tensors and an IDENTICAL removed list for each; load_network
    the same through a file."""
    import json
    import os
    import numpy as np
    from inputs import ckpt_cases, ckpt_toy_net
    gold = np.load(os.path.join(golden_dir, "load_network_cases.npz"))
    meta = json.load(open(os.path.join(golden_dir, "load_network_cases.json")))
    cases = ckpt_cases()
    assert set(cases) == set(meta)
    for name, ckpt in cases.items():
        net = ckpt_toy_net(0)
        out, removed = remap_state_dict(net.state_dict(), ckpt)
        assert removed == meta[name], (name, removed, meta[name])
        for k, v in out.items():
            assert np.array_equal(v.numpy(), gold[f"{name}/{k}"]), (name, k)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "save_step_7.pth")
        torch.save(cases["mismatch_prefixed"], path)
        net, removed = load_network(ckpt_toy_net(0), path)
        assert removed == meta["mismatch_prefixed"]
        for k, v in net.state_dict().items():
            assert np.array_equal(v.numpy(), gold[f"mismatch_prefixed/{k}"]), k


def test_select_checkpoint_follows_the_evaluator(tmp_path):
    """networks/managers/evaluator.py:59-110: 'test' loads nothing; an explicit path is taken as is; with no path the
    explicit step or the LARGEST step in DIR_CKPT is used, and TEST_EMA switches the directory to <DIR_RESULT>/ema_ckpt
    after the step was chosen from the plain directory's listing."""
    import types
    from rmem_amd.checkpoint import load_for_evaluation, select_checkpoint
    from inputs import ckpt_toy_net
    res = tmp_path / "result"
    ck, ema = res / "ckpt", res / "ema_ckpt"
    ck.mkdir(parents=True), ema.mkdir()
    src = ckpt_toy_net(1)
    for step in (900, 12000, 3000):
        torch.save({"state_dict": src.state_dict(), "optimizer": {}}, ck / f"save_step_{step}.pth")     # save_network's format (:104-121)
    ema_sd = {k: v * 2 for k, v in src.state_dict().items()}
    torch.save({"state_dict": ema_sd}, ema / "save_step_12000.pth")
    mk = lambda **kw: types.SimpleNamespace(**{**dict(TEST_CKPT_PATH=None, TEST_CKPT_STEP=None, TEST_EMA=False,
                                                      DIR_CKPT=str(ck), DIR_RESULT=str(res)), **kw})
    assert select_checkpoint(mk(TEST_CKPT_PATH="test")) == ("test", None)
    assert select_checkpoint(mk(TEST_CKPT_PATH="/x/y.pth")) == ("unknown", "/x/y.pth")
    c = mk()
    assert select_checkpoint(c) == ("12000", str(ck / "save_step_12000.pth")) and c.TEST_CKPT_PATH.endswith("save_step_12000.pth")
    assert select_checkpoint(mk(TEST_CKPT_STEP=3000)) == ("3000", str(ck / "save_step_3000.pth"))
    c = mk(TEST_EMA=True)
    assert select_checkpoint(c) == ("12000", str(ema / "save_step_12000.pth")) and c.DIR_CKPT == str(ema)
    empty = tmp_path / "none"
    empty.mkdir()
    import pytest
    with pytest.raises(FileNotFoundError):
        select_checkpoint(mk(DIR_CKPT=str(empty)))
    net, label, removed = load_for_evaluation(ckpt_toy_net(0), mk(TEST_EMA=True))
    assert label == "12000" and removed == []
    assert all(torch.equal(v, ema_sd[k]) for k, v in net.state_dict().items())
