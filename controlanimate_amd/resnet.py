"""ResnetBlock3D / Upsample3D / Downsample3D on the HIP kernels (NHWC, frames folded into batch).

Mirrors the module names and checkpoint keys of the reference's animatediff/models/resnet.py
(InflatedConv3d :12-20, InflatedGroupNorm :23-31, Upsample3D :34-82, Downsample3D :85-108,
ResnetBlock3D :111-218).  "Inflation" needs no code here: activations are stored as
[b*f, h, w, c], so a per-frame conv IS a 2-D NHWC conv over b*f images.  Fusions vs the reference:
  conv1 epilogue: + bias + time_emb_proj(silu(temb)) broadcast   (reference :196-200)
  conv2 epilogue: (+ shortcut/input) * 1/output_scale_factor       (:216)
  Upsample3D:     nearest x2 folded into the conv's input gather   (:67 + :81)
  skip concat:    GroupNorm / shortcut read the two tensors directly (unet_blocks.py:636,742)
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import kernels as K
from .context import ExecCtx
from .layers import HipConv1x1, HipConv3x3, HipGroupNorm, HipLinear, WeightArena


class InflatedConv3d(HipConv3x3):
    """3x3 (stride 1/2) per-frame convolution; parameters as nn.Conv2d."""


class InflatedGroupNorm(HipGroupNorm):
    """GroupNorm with per-frame statistics (v2) -- or cross-frame when the model says so (v1)."""


class Upsample3D(nn.Module):
    def __init__(self, channels, use_conv=True, out_channels=None, name="conv"):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels or channels
        self.conv = InflatedConv3d(channels, self.out_channels)

    def pack(self, arena, dtype):
        self.conv.pack(arena, dtype)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.conv.run(x, upsample=True)


class Downsample3D(nn.Module):
    def __init__(self, channels, use_conv=True, out_channels=None, padding=1, name="conv"):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels or channels
        self.conv = InflatedConv3d(channels, self.out_channels, stride=2)

    def pack(self, arena, dtype):
        self.conv.pack(arena, dtype)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.conv.run(x)


class ResnetBlock3D(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, temb_channels=512, groups=32, eps=1e-6,
                 output_scale_factor=1.0, use_inflated_groupnorm=True, **_):
        super().__init__()
        out_channels = out_channels or in_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.output_scale_factor = output_scale_factor
        self.use_inflated_groupnorm = use_inflated_groupnorm
        self.norm1 = InflatedGroupNorm(groups, in_channels, eps)
        self.conv1 = InflatedConv3d(in_channels, out_channels)
        self.time_emb_proj = HipLinear(temb_channels, out_channels)
        self.norm2 = InflatedGroupNorm(groups, out_channels, eps)
        self.conv2 = InflatedConv3d(out_channels, out_channels)
        self.conv_shortcut = HipConv1x1(in_channels, out_channels) if in_channels != out_channels else None
        self.temb_slice = (0, 0)  # column range inside ExecCtx.temb, assigned by the owning model

    def pack(self, arena: WeightArena, dtype):
        for m in (self.norm1, self.conv1, self.norm2, self.conv2):
            m.pack(arena, dtype)
        if self.conv_shortcut is not None:
            self.conv_shortcut.pack(arena, dtype)
        # time_emb_proj is packed by the model into one fused [sum C_out, temb] matrix

    def forward(self, x: torch.Tensor, ctx: ExecCtx, skip: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x [images,h,w,c1] (+ skip [images,h,w,c2] standing for torch.cat([x, skip], dim=1))."""
        images, h, w, _ = x.shape
        lo, hi = self.temb_slice
        kw1 = dict(rowbias=ctx.temb[:, lo:hi], rows_per_group=ctx.rows_per_emb_group(h, w))
        h1 = self._norm_conv(self.norm1, self.conv1, x, skip, ctx, **kw1)
        if self.conv_shortcut is not None:
            rows = images * h * w
            res = self.conv_shortcut.run(x.view(rows, x.shape[3]), a2=None if skip is None else skip.view(rows, skip.shape[3]))
        else:
            assert skip is None
            res = x
        return self._norm_conv(self.norm2, self.conv2, h1, None, ctx, residual=res.view(images, h, w, self.out_channels),
                               post_scale=1.0 / self.output_scale_factor)

    @staticmethod
    def _norm_conv(norm, conv, x, skip, ctx: ExecCtx, **conv_kw) -> torch.Tensor:
        """conv(silu(norm(cat(x, skip)))) (reference :188-212).  Where the convolution takes the Winograd route and the one-launch
        GroupNorm applies (per-frame statistics, the 8x8- / 16x16-latent levels) the GroupNorm writes the convolution's transformed
        input itself: K.group_norm_conv3x3_wino."""
        u = getattr(conv, "u", None)
        if u is not None and ctx.gn_frames_per_stat == 1:
            y = K.group_norm_conv3x3_wino(x, norm.g.t, norm.b.t, conv.w.t, u.t, x2=skip, groups=norm.num_groups, eps=norm.eps, act=K.ACT_SILU,
                                          bias=conv.b.t, **conv_kw)
            if y is not None:
                return y
        h = norm.run(x, x2=skip, frames_per_stat=ctx.gn_frames_per_stat, act=K.ACT_SILU)
        return conv.run(h, **conv_kw)
