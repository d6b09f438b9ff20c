"""FPN segmentation head, key-compatible with the reference decoder
(/root/reference/aot_plus/networks/decoders/fpn.py:7-68, layers/basic.py:60-70).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class _ConvGN(nn.Module):
    def __init__(self, cin, cout, k, groups=8):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=k // 2)
        self.gn = nn.GroupNorm(groups, cout)

    def forward(self, x, relu=False):
        if x.is_cuda and not torch.is_grad_enabled():
            # bias-free convolution; its bias, the GroupNorm and the ReLU in two HIP passes
            from ..hip import groupnorm_nchw
            c = self.conv
            y = F.conv2d(x, c.weight, None, c.stride, c.padding, c.dilation, c.groups)
            return groupnorm_nchw(y, self.gn, relu, conv_bias=c.bias)
        y = self.conv(x)
        y = self.gn(y)
        return F.relu(y) if relu else y


class FPNHead(nn.Module):
    def __init__(self, in_dim, out_dim, hidden_dim=256,
                 shortcut_dims=(256, 512, 1024, 1024), align_corners=True,
                 decode_intermediate_input=False):
        super().__init__()
        self.align_corners = align_corners
        self.decode_intermediate_input = decode_intermediate_input
        h = hidden_dim
        self.conv_in = _ConvGN(in_dim, h, 1)
        self.conv_16x = _ConvGN(h, h, 3)
        self.conv_8x = _ConvGN(h, h // 2, 3)
        self.conv_4x = _ConvGN(h // 2, h // 2, 3)
        self.adapter_16x = nn.Conv2d(shortcut_dims[-2], h, 1)
        self.adapter_8x = nn.Conv2d(shortcut_dims[-3], h, 1)
        self.adapter_4x = nn.Conv2d(shortcut_dims[-4], h // 2, 1)
        self.conv_out = nn.Conv2d(h // 2, out_dim, 1)

    def _up(self, x, like):
        return F.interpolate(x, size=like.shape[-2:], mode="bilinear",
                             align_corners=self.align_corners)

    def adapter_convs(self, shortcuts):
        """The bias-free 1x1 adapter convolutions of the three skip connections.  They depend on
        the encoder features only, so the engine runs them with the encoder pass (prefetched on
        the encoder stream) instead of on the frame's critical path between LSTT and labels."""
        return [F.conv2d(shortcuts[i], a.weight, None, a.stride, a.padding)
                for i, a in ((-2, self.adapter_16x), (-3, self.adapter_8x), (-4, self.adapter_4x))]

    def _merge(self, adapter: nn.Conv2d, skip, x, pre=None):
        """adapter(skip) + upsample(x): on the GPU the adapter's bias, the bilinear upsample and the
        sum are one HIP pass over the adapter's conv output (rmem_upsample_add_nchw).  `pre`: that
        conv output if it was computed with the encoder pass; it is NOT modified (a prefetched
        feature set may be decoded more than once)."""
        if skip.is_cuda and not torch.is_grad_enabled():
            from ..hip import upsample_add_nchw_
            if pre is not None:
                return upsample_add_nchw_(pre, adapter.bias, x, self.align_corners, inplace=False)
            y = F.conv2d(skip, adapter.weight, None, adapter.stride, adapter.padding)
            return upsample_add_nchw_(y, adapter.bias, x, self.align_corners)
        return adapter(skip) + (x if x.shape[-2:] == skip.shape[-2:] else self._up(x, skip))

    def forward(self, inputs, shortcuts):
        x = torch.cat(inputs, dim=1) if self.decode_intermediate_input else inputs[-1]
        x = self.conv_in(x, relu=True)
        if x.is_cuda and not torch.is_grad_enabled():
            pre = getattr(shortcuts, "adapters", None) or (None, None, None)
            x = self.conv_16x(self._merge(self.adapter_16x, shortcuts[-2], x, pre[0]), relu=True)
            x = self.conv_8x(self._merge(self.adapter_8x, shortcuts[-3], x, pre[1]), relu=True)
            x = self.conv_4x(self._merge(self.adapter_4x, shortcuts[-4], x, pre[2]), relu=True)
            return self.conv_out(x)
        x = self.conv_16x(self.adapter_16x(shortcuts[-2]) + x, relu=True)
        x = self.conv_8x(self.adapter_8x(shortcuts[-3]) + self._up(x, shortcuts[-3]), relu=True)
        x = self.conv_4x(self.adapter_4x(shortcuts[-4]) + self._up(x, shortcuts[-4]), relu=True)
        return self.conv_out(x)
