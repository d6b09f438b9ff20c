"""rmem_amd -- MI355X-native RMem (restricted-memory AOT/DeAOT) inference hot path."""
from .config import get_config, ModelConfig  # noqa: F401

__version__ = "0.1.0"
