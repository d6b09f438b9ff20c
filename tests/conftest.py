import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)


# The suite runs with reproducible MIOpen convolutions (rmem_amd/determinism.py: the implicit-GEMM solver family,
# whose output varies from call to call at the small test geometries, is switched off before the first
# convolution of the process -- spawned workers inherit the environment).
os.environ.setdefault("MIOPEN_DEBUG_CONV_IMPLICIT_GEMM", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def deaot_model():
    """Our model container with the name-keyed synthetic weights (CPU)."""
    import torch
    from rmem_amd.config import get_config
    from rmem_amd.model import build_vos_model
    from rmem_amd.synth import load_synthetic_weights
    torch.manual_seed(0)
    m = build_vos_model("deaot", get_config("r50_deaotl")).eval()
    load_synthetic_weights(m)
    return m
