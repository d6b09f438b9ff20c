"""ResNet-50 trunk (output stride 16, last stage dropped) with frozen BatchNorm.

Key-compatible with the reference encoder's ``state_dict``
(/root/reference/aot_plus/networks/encoders/resnet.py:71-196,359-374 and
 networks/layers/normalization.py:6-43): ``conv1``, ``bn1``, ``layer{1,2,3}.{i}.
{conv1,bn1,conv2,bn2,conv3,bn3,downsample.0,downsample.1}``.  Returns the
pyramid [4x, 8x, 16x, 16x].
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class FrozenBN(nn.Module):
    """Inference-only affine BN with buffers (normalization.py:6-43)."""

    def __init__(self, n: int, eps: float = 1e-5):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n) - eps)
        self.eps = eps

    def forward(self, x):
        return F.batch_norm(x, self.running_mean, self.running_var, self.weight,
                            self.bias, training=False, eps=self.eps)


def _fold(conv: nn.Conv2d, bn: "FrozenBN") -> nn.Conv2d:
    """conv followed by a frozen BN -> one conv with bias (inference-time folding)."""
    scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
    out = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding,
                    conv.dilation, conv.groups, bias=True).to(conv.weight.device)
    with torch.no_grad():
        out.weight.copy_(conv.weight * scale.view(-1, 1, 1, 1))
        out.bias.copy_(bn.bias - bn.running_mean * scale)
    return out


class _Bottleneck(nn.Module):
    def __init__(self, cin, width, stride=1, dilation=1, proj=False):
        super().__init__()
        cout = width * 4
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = FrozenBN(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=dilation,
                               dilation=dilation, bias=False)
        self.bn2 = FrozenBN(width)
        self.conv3 = nn.Conv2d(width, cout, 1, bias=False)
        self.bn3 = FrozenBN(cout)
        self.downsample = None
        if proj:
            self.downsample = nn.Sequential(
                nn.Conv2d(cin, cout, 1, stride=stride, bias=False), FrozenBN(cout))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        s = x if self.downsample is None else self.downsample(x)
        return F.relu(y + s)


def _stage(cin, width, n, stride, dilation):
    # first block dilation = max(dilation // 2, 1) (resnet.py:160-165)
    blocks = [_Bottleneck(cin, width, stride, max(dilation // 2, 1),
                          proj=(stride != 1 or cin != width * 4))]
    blocks += [_Bottleneck(width * 4, width, 1, dilation) for _ in range(n - 1)]
    return nn.Sequential(*blocks)


class ResNet50Encoder(nn.Module):
    out_dims = (256, 512, 1024, 1024)

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = FrozenBN(64)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = _stage(64, 64, 3, 1, 1)
        self.layer2 = _stage(256, 128, 4, 2, 1)
        self.layer3 = _stage(512, 256, 6, 2, 1)

    def folded(self) -> "ResNet50Encoder":
        """Inference copy with every FrozenBN folded into the preceding conv (same function up
        to fp32 rounding; removes 43 BatchNorm launches per frame).  Keys no longer match the
        reference's state_dict, so this is built from the loaded model, never loaded into."""
        import copy
        m = copy.deepcopy(self)
        m.conv1, m.bn1 = _fold(m.conv1, m.bn1), nn.Identity()
        for stage in (m.layer1, m.layer2, m.layer3):
            for blk in stage:
                blk.conv1, blk.bn1 = _fold(blk.conv1, blk.bn1), nn.Identity()
                blk.conv2, blk.bn2 = _fold(blk.conv2, blk.bn2), nn.Identity()
                blk.conv3, blk.bn3 = _fold(blk.conv3, blk.bn3), nn.Identity()
                if blk.downsample is not None:
                    blk.downsample = nn.Sequential(_fold(blk.downsample[0], blk.downsample[1]))
        return m.eval()

    def forward(self, img):
        x = self.maxpool(F.relu(self.bn1(self.conv1(img))))
        c4 = self.layer1(x)
        c8 = self.layer2(c4)
        c16 = self.layer3(c8)
        return [c4, c8, c16, c16]
