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


class _FoldedConv(nn.Module):
    """conv + frozen BN folded into (bias-free conv, per-channel bias); the bias, the optional
    residual and the ReLU are applied by one HIP pass on the GPU (rmem_bias_act_nchw)."""

    def __init__(self, conv: nn.Conv2d, bn: "FrozenBN"):
        super().__init__()
        scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
        self.conv = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding,
                              conv.dilation, conv.groups, bias=False).to(conv.weight.device)
        with torch.no_grad():
            self.conv.weight.copy_(conv.weight * scale.view(-1, 1, 1, 1))
        self.register_buffer("bias", (bn.bias - bn.running_mean * scale).detach().clone())

    def forward(self, x, residual=None, relu=True):
        y = self.conv(x)
        if y.is_cuda and not torch.is_grad_enabled():
            from ..hip import bias_act_nchw_
            return bias_act_nchw_(y.contiguous(), self.bias, residual, relu)
        y = y + self.bias.view(1, -1, 1, 1)
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y


class _FoldedBottleneck(nn.Module):
    def __init__(self, b: "_Bottleneck"):
        super().__init__()
        self.c1, self.c2, self.c3 = _FoldedConv(b.conv1, b.bn1), _FoldedConv(b.conv2, b.bn2), _FoldedConv(b.conv3, b.bn3)
        self.down = _FoldedConv(b.downsample[0], b.downsample[1]) if b.downsample is not None else None

    def forward(self, x):
        s = x if self.down is None else self.down(x, relu=False)
        y = self.c2(self.c1(x))
        return self.c3(y, residual=s.contiguous(), relu=True)


class _FoldedResNet(nn.Module):
    def __init__(self, m: "ResNet50Encoder"):
        super().__init__()
        self.stem = _FoldedConv(m.conv1, m.bn1)
        self.maxpool = m.maxpool
        self.layer1 = nn.Sequential(*[_FoldedBottleneck(b) for b in m.layer1])
        self.layer2 = nn.Sequential(*[_FoldedBottleneck(b) for b in m.layer2])
        self.layer3 = nn.Sequential(*[_FoldedBottleneck(b) for b in m.layer3])

    def forward(self, img):
        x = self.maxpool(self.stem(img))
        c4 = self.layer1(x)
        c8 = self.layer2(c4)
        c16 = self.layer3(c8)
        return [c4, c8, c16, c16]


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

    def folded(self) -> nn.Module:
        """Inference copy with every FrozenBN folded into the preceding conv and the bias /
        residual / ReLU epilogues fused (same function up to fp32 rounding; removes ~110
        pointwise launches per frame).  Built from the loaded model, never loaded into."""
        return _FoldedResNet(self).eval()

    def forward(self, img):
        x = self.maxpool(F.relu(self.bn1(self.conv1(img))))
        c4 = self.layer1(x)
        c8 = self.layer2(c4)
        c16 = self.layer3(c8)
        return [c4, c8, c16, c16]
