"""Seeded inputs shared by make_golden.py (generator) and the tests (consumers)."""
import torch


def tta_new_object_label(out_hw):
    """Label map [1,1,H0,W0] introducing object id 4 (a rectangle) mid-clip."""
    lab = torch.zeros(1, 1, *out_hw)
    lab[:, :, out_hw[0] // 8: out_hw[0] // 3, out_hw[1] // 2: out_hw[1] * 3 // 4] = 4
    return lab
