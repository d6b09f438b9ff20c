"""Plain-PyTorch encoder / decoder definitions (run through PyTorch-ROCm/MIOpen).

These sit *outside* the HIP hot path: they exist so
that a reference checkpoint's ``state_dict`` loads by name.
"""
from .resnet import ResNet50Encoder, FrozenBN
from .fpn import FPNHead
from .swin import SwinBEncoder

__all__ = ["ResNet50Encoder", "FrozenBN", "FPNHead", "SwinBEncoder"]
