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
