"""Swin-B trunk (3 stages: strides 4 / 8 / 16, dims 128 / 256 / 512, depths 2 / 2 / 18, window 7),
plain PyTorch, key-compatible with the reference encoder
(/root/reference/aot_plus/networks/encoders/swin/swin_transformer.py:521-716 with the
``swin_base`` arguments of swin/build.py:10-22).  Runs through PyTorch-ROCm unchanged; it
is outside the HIP hot path (BASELINE.json configs[4]).  Returns [4x, 8x, 16x, 16x].
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


def _windows(x, ws):
    """[B,H,W,C] -> [B*nW, ws*ws, C]"""
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, ws * ws, C)


def _unwindows(wins, ws, H, W):
    B = wins.shape[0] // ((H // ws) * (W // ws))
    x = wins.view(B, H // ws, W // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(B, H, W, -1)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class _WindowAttention(nn.Module):
    def __init__(self, dim, ws, heads):
        super().__init__()
        self.ws, self.heads = ws, heads
        self.scale = (dim // heads) ** -0.5
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, heads))
        c = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
        rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0) + (ws - 1)
        self.register_buffer("relative_position_index", rel[:, :, 0] * (2 * ws - 1) + rel[:, :, 1])
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x, mask=None):
        B_, N, C = x.shape
        qkv = self.qkv(x).reshape(B_, N, 3, self.heads, C // self.heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * self.scale, qkv[1], qkv[2]
        attn = q @ k.transpose(-2, -1)
        bias = self.relative_position_bias_table[self.relative_position_index.view(-1)].view(N, N, -1)
        attn = attn + bias.permute(2, 0, 1).unsqueeze(0)
        if mask is not None:
            nW = mask.shape[0]
            attn = (attn.view(B_ // nW, nW, self.heads, N, N) + mask[None, :, None]).view(-1, self.heads, N, N)
        attn = torch.softmax(attn, dim=-1)
        return self.proj((attn @ v).transpose(1, 2).reshape(B_, N, C))


class _Block(nn.Module):
    def __init__(self, dim, heads, ws, shift):
        super().__init__()
        self.ws, self.shift = ws, shift
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _WindowAttention(dim, ws, heads)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _Mlp(dim, dim * 4)

    def forward(self, x, H, W, mask):
        B, L, C = x.shape
        ws = self.ws
        y = self.norm1(x).view(B, H, W, C)
        pr, pb = (ws - W % ws) % ws, (ws - H % ws) % ws
        y = F.pad(y, (0, 0, 0, pr, 0, pb))
        Hp, Wp = H + pb, W + pr
        if self.shift > 0:
            y = torch.roll(y, shifts=(-self.shift, -self.shift), dims=(1, 2))
        y = self.attn(_windows(y, ws), mask if self.shift > 0 else None)
        y = _unwindows(y, ws, Hp, Wp)
        if self.shift > 0:
            y = torch.roll(y, shifts=(self.shift, self.shift), dims=(1, 2))
        y = y[:, :H, :W, :].reshape(B, H * W, C)
        x = x + y
        return x + self.mlp(self.norm2(x))


class _Merge(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)

    def forward(self, x, H, W):
        B, L, C = x.shape
        x = x.view(B, H, W, C)
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
        return self.reduction(self.norm(x.view(B, -1, 4 * C)))


class _Stage(nn.Module):
    def __init__(self, dim, depth, heads, ws, merge):
        super().__init__()
        self.ws, self.shift = ws, ws // 2
        self.blocks = nn.ModuleList(_Block(dim, heads, ws, 0 if i % 2 == 0 else ws // 2) for i in range(depth))
        self.downsample = _Merge(dim) if merge else None

    def _mask(self, H, W, device):
        ws, sh = self.ws, self.shift
        Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
        img = torch.zeros(1, Hp, Wp, 1, device=device)
        cnt = 0
        for hs in (slice(0, -ws), slice(-ws, -sh), slice(-sh, None)):
            for wsl in (slice(0, -ws), slice(-ws, -sh), slice(-sh, None)):
                img[:, hs, wsl, :] = cnt
                cnt += 1
        mw = _windows(img, ws).view(-1, ws * ws)
        m = mw.unsqueeze(1) - mw.unsqueeze(2)
        return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)

    def forward(self, x, H, W):
        mask = self._mask(H, W, x.device)
        for blk in self.blocks:
            x = blk(x, H, W, mask)
        if self.downsample is not None:
            return x, self.downsample(x, H, W), (H + 1) // 2, (W + 1) // 2
        return x, x, H, W


class _PatchEmbed(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, 4, stride=4)
        self.norm = nn.LayerNorm(dim)

    def forward(self, x):
        _, _, H, W = x.shape
        x = F.pad(x, (0, (4 - W % 4) % 4, 0, (4 - H % 4) % 4))
        x = self.proj(x)
        Wh, Ww = x.shape[2:]
        x = self.norm(x.flatten(2).transpose(1, 2))
        return x, Wh, Ww


class SwinBEncoder(nn.Module):
    out_dims = (128, 256, 512, 512)

    def __init__(self, embed_dim=128, depths=(2, 2, 18), heads=(4, 8, 16), ws=7):
        super().__init__()
        self.patch_embed = _PatchEmbed(embed_dim)
        self.layers = nn.ModuleList(
            _Stage(embed_dim * 2 ** i, depths[i], heads[i], ws, merge=(i < len(depths) - 1))
            for i in range(len(depths)))
        for i in range(len(depths)):
            self.add_module(f"norm{i}", nn.LayerNorm(embed_dim * 2 ** i))

    def folded(self):
        return None     # nothing to fold; present so that optimize_for_inference() is uniform

    def forward(self, img):
        x, H, W = self.patch_embed(img)
        outs = []
        for i, stage in enumerate(self.layers):
            x_out, x, Hn, Wn = stage(x, H, W)
            y = getattr(self, f"norm{i}")(x_out)
            outs.append(y.view(-1, H, W, y.shape[-1]).permute(0, 3, 1, 2).contiguous())
            H, W = Hn, Wn
        outs.append(outs[-1])
        return outs
