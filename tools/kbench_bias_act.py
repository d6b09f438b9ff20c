"""rmem_bias_act_nchw[_batched] on the encoder's feature-map shapes (batch 2 = the prefetched pass): us per launch and the
bandwidth over the bytes the launch must move (x read + written, residual read).  python tools/kbench_bias_act.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rmem_amd.hip import bias_act_nchw_   # noqa: E402

dev = "cuda:0"
shapes = [(2, 64, 241, 425, False), (2, 64, 121, 213, False), (2, 256, 121, 213, True), (2, 128, 121, 213, False),
          (2, 128, 61, 107, False), (2, 512, 61, 107, True), (2, 256, 61, 107, False), (2, 256, 31, 54, False),
          (2, 1024, 31, 54, True), (1, 256, 121, 213, False)]
for B, C, H, W, res in shapes:
    x = torch.randn(B, C, H, W, device=dev)
    r = torch.randn(B, C, H, W, device=dev) if res else None
    b = torch.randn(C, device=dev)
    for _ in range(5):
        bias_act_nchw_(x, b, r, True)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            bias_act_nchw_(x, b, r, True)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 100 * 1e3
    nbytes = x.numel() * 4 * (3 if res else 2)
    print(f"{B}x{C}x{H}x{W} residual={res}: {us:6.2f} us  {nbytes / 1e6:6.1f} MB  {nbytes / us / 1e6:5.2f} TB/s", flush=True)
