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
    # rule 2 is checked before the 'module.' retry in the reference, so a prefixed 11-channel
    # bank is *not* zero-extended there either: it falls through to "removed"
    assert "module.patch_wise_id_bank.weight" in removed
    assert torch.equal(new["decoder.conv_out.weight"], before_out)
    assert set(removed) == {"module.patch_wise_id_bank.weight", "module.decoder.conv_out.weight", "optimizer_junk"}
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
    m.optimize_for_inference(True, force=True)
    assert m.__dict__["_enc_infer"] is not second and m.__dict__["_weights_version"] == v0 + 2
    m.optimize_for_inference(False)
    assert m.__dict__["_enc_infer"] is None
    assert "_enc_infer_state" not in m.state_dict()
