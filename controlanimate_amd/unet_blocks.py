"""Block containers of the UNet3D (and, with motion modules off, of the ControlNet encoder).

Same class names, constructor arguments, sub-module names (`resnets`, `attentions`,
`motion_modules`, `downsamplers`, `upsamplers`) and per-layer order resnet -> spatial transformer
-> motion module as the reference's animatediff/models/unet_blocks.py (:173-280, :283-423, :426-523,
:526-669, :672-762).  The up-blocks never materialise torch.cat([hidden, skip]): the resnet's
GroupNorm and shortcut read both tensors (see resnet.py).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
from torch import nn

from .attention import Transformer3DModel
from .context import ExecCtx
from .motion_module import get_motion_module
from .resnet import Downsample3D, ResnetBlock3D, Upsample3D


def _resnet(cin, cout, temb, eps, groups, scale, infl):
    return ResnetBlock3D(in_channels=cin, out_channels=cout, temb_channels=temb, eps=eps, groups=groups,
                         output_scale_factor=scale, use_inflated_groupnorm=infl)


def _transformer(heads, channels, cross_dim, groups, kw):
    return Transformer3DModel(heads, channels // heads, in_channels=channels, num_layers=1, cross_attention_dim=cross_dim,
                              norm_num_groups=groups, **kw)


def _motion(channels, use, mtype, mkwargs):
    return get_motion_module(in_channels=channels, motion_module_type=mtype, motion_module_kwargs=mkwargs) if use else None


class _Block(nn.Module):
    def pack(self, arena, dtype):
        for name in ("resnets", "attentions", "motion_modules", "downsamplers", "upsamplers"):
            mods = getattr(self, name, None)
            if mods is None:
                continue
            for m in mods:
                if m is not None:
                    m.pack(arena, dtype)


class CrossAttnDownBlock3D(_Block):
    has_cross_attention = True

    def __init__(self, in_channels, out_channels, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32,
                 attn_num_head_channels=1, cross_attention_dim=1280, output_scale_factor=1.0, add_downsample=True,
                 use_inflated_groupnorm=None, use_motion_module=None, motion_module_type=None, motion_module_kwargs=None,
                 unet_use_cross_frame_attention=False, unet_use_temporal_attention=False, **_):
        super().__init__()
        kw = dict(unet_use_cross_frame_attention=unet_use_cross_frame_attention, unet_use_temporal_attention=unet_use_temporal_attention)
        self.resnets = nn.ModuleList([_resnet(in_channels if i == 0 else out_channels, out_channels, temb_channels, resnet_eps,
                                              resnet_groups, output_scale_factor, use_inflated_groupnorm) for i in range(num_layers)])
        self.attentions = nn.ModuleList([_transformer(attn_num_head_channels, out_channels, cross_attention_dim, resnet_groups, kw)
                                         for _ in range(num_layers)])
        self.motion_modules = nn.ModuleList([_motion(out_channels, use_motion_module, motion_module_type, motion_module_kwargs)
                                             for _ in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample3D(out_channels, use_conv=True, out_channels=out_channels)]) if add_downsample else None

    def forward(self, x: torch.Tensor, ctx: ExecCtx, half_ctx: Optional[ExecCtx] = None) -> Tuple[torch.Tensor, List[torch.Tensor]]:
        """half_ctx: `x` is one of the two identical CFG halves (see Transformer3DModel.forward): the first resnet and the
        prompt-independent head of the first transformer run on it, everything after on the full batch."""
        outs = []
        for i, (resnet, attn, mm) in enumerate(zip(self.resnets, self.attentions, self.motion_modules)):
            if i == 0 and half_ctx is not None:
                x = attn(resnet(x, half_ctx), ctx, shared_half=True)
            else:
                x = attn(resnet(x, ctx), ctx)
            if mm is not None:
                x = mm(x, ctx)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class DownBlock3D(_Block):
    has_cross_attention = False

    def __init__(self, in_channels, out_channels, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32,
                 output_scale_factor=1.0, add_downsample=True, use_inflated_groupnorm=None, use_motion_module=None,
                 motion_module_type=None, motion_module_kwargs=None, **_):
        super().__init__()
        self.resnets = nn.ModuleList([_resnet(in_channels if i == 0 else out_channels, out_channels, temb_channels, resnet_eps,
                                              resnet_groups, output_scale_factor, use_inflated_groupnorm) for i in range(num_layers)])
        self.motion_modules = nn.ModuleList([_motion(out_channels, use_motion_module, motion_module_type, motion_module_kwargs)
                                             for _ in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample3D(out_channels, use_conv=True, out_channels=out_channels)]) if add_downsample else None

    def forward(self, x, ctx):
        outs = []
        for resnet, mm in zip(self.resnets, self.motion_modules):
            x = resnet(x, ctx)
            if mm is not None:
                x = mm(x, ctx)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class UNetMidBlock3DCrossAttn(_Block):
    has_cross_attention = True

    def __init__(self, in_channels, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32, attn_num_head_channels=1,
                 output_scale_factor=1.0, cross_attention_dim=1280, use_inflated_groupnorm=None, use_motion_module=None,
                 motion_module_type=None, motion_module_kwargs=None, unet_use_cross_frame_attention=False,
                 unet_use_temporal_attention=False, **_):
        super().__init__()
        kw = dict(unet_use_cross_frame_attention=unet_use_cross_frame_attention, unet_use_temporal_attention=unet_use_temporal_attention)
        resnets = [_resnet(in_channels, in_channels, temb_channels, resnet_eps, resnet_groups, output_scale_factor, use_inflated_groupnorm)]
        attentions, motion_modules = [], []
        for _i in range(num_layers):
            attentions.append(_transformer(attn_num_head_channels, in_channels, cross_attention_dim, resnet_groups, kw))
            motion_modules.append(_motion(in_channels, use_motion_module, motion_module_type, motion_module_kwargs))
            resnets.append(_resnet(in_channels, in_channels, temb_channels, resnet_eps, resnet_groups, output_scale_factor, use_inflated_groupnorm))
        self.attentions = nn.ModuleList(attentions)
        self.resnets = nn.ModuleList(resnets)
        self.motion_modules = nn.ModuleList(motion_modules)

    def forward(self, x, ctx):
        x = self.resnets[0](x, ctx)
        for attn, resnet, mm in zip(self.attentions, self.resnets[1:], self.motion_modules):
            x = attn(x, ctx)
            if mm is not None:
                x = mm(x, ctx)
            x = resnet(x, ctx)
        return x


class CrossAttnUpBlock3D(_Block):
    has_cross_attention = True

    def __init__(self, in_channels, out_channels, prev_output_channel, temb_channels, num_layers=1, resnet_eps=1e-6,
                 resnet_groups=32, attn_num_head_channels=1, cross_attention_dim=1280, output_scale_factor=1.0,
                 add_upsample=True, use_inflated_groupnorm=None, use_motion_module=None, motion_module_type=None,
                 motion_module_kwargs=None, unet_use_cross_frame_attention=False, unet_use_temporal_attention=False, **_):
        super().__init__()
        kw = dict(unet_use_cross_frame_attention=unet_use_cross_frame_attention, unet_use_temporal_attention=unet_use_temporal_attention)
        resnets = []
        for i in range(num_layers):
            skip = in_channels if i == num_layers - 1 else out_channels
            rin = prev_output_channel if i == 0 else out_channels
            resnets.append(_resnet(rin + skip, out_channels, temb_channels, resnet_eps, resnet_groups, output_scale_factor, use_inflated_groupnorm))
        self.resnets = nn.ModuleList(resnets)
        self.attentions = nn.ModuleList([_transformer(attn_num_head_channels, out_channels, cross_attention_dim, resnet_groups, kw)
                                         for _ in range(num_layers)])
        self.motion_modules = nn.ModuleList([_motion(out_channels, use_motion_module, motion_module_type, motion_module_kwargs)
                                             for _ in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample3D(out_channels, use_conv=True, out_channels=out_channels)]) if add_upsample else None

    def forward(self, x, skips: Sequence[torch.Tensor], ctx):
        skips = list(skips)
        for resnet, attn, mm in zip(self.resnets, self.attentions, self.motion_modules):
            x = resnet(x, ctx, skip=skips.pop())
            x = attn(x, ctx)
            if mm is not None:
                x = mm(x, ctx)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class UpBlock3D(_Block):
    has_cross_attention = False

    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers=1, resnet_eps=1e-6,
                 resnet_groups=32, output_scale_factor=1.0, add_upsample=True, use_inflated_groupnorm=None,
                 use_motion_module=None, motion_module_type=None, motion_module_kwargs=None, **_):
        super().__init__()
        resnets = []
        for i in range(num_layers):
            skip = in_channels if i == num_layers - 1 else out_channels
            rin = prev_output_channel if i == 0 else out_channels
            resnets.append(_resnet(rin + skip, out_channels, temb_channels, resnet_eps, resnet_groups, output_scale_factor, use_inflated_groupnorm))
        self.resnets = nn.ModuleList(resnets)
        self.motion_modules = nn.ModuleList([_motion(out_channels, use_motion_module, motion_module_type, motion_module_kwargs)
                                             for _ in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample3D(out_channels, use_conv=True, out_channels=out_channels)]) if add_upsample else None

    def forward(self, x, skips, ctx):
        skips = list(skips)
        for resnet, mm in zip(self.resnets, self.motion_modules):
            x = resnet(x, ctx, skip=skips.pop())
            if mm is not None:
                x = mm(x, ctx)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


def get_down_block(down_block_type, **kw):
    t = down_block_type[7:] if down_block_type.startswith("UNetRes") else down_block_type
    if t in ("DownBlock3D", "DownBlock2D"):
        return DownBlock3D(**kw)
    if t in ("CrossAttnDownBlock3D", "CrossAttnDownBlock2D"):
        return CrossAttnDownBlock3D(**kw)
    raise ValueError(f"{down_block_type} does not exist.")


def get_up_block(up_block_type, **kw):
    t = up_block_type[7:] if up_block_type.startswith("UNetRes") else up_block_type
    if t == "UpBlock3D":
        return UpBlock3D(**kw)
    if t == "CrossAttnUpBlock3D":
        return CrossAttnUpBlock3D(**kw)
    raise ValueError(f"{up_block_type} does not exist.")
