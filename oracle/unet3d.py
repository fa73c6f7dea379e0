"""fp32 oracle of the reference's UNet3DConditionModel (AnimateDiff-inflated SD1.5 UNet).

Functional over a flat state dict that uses the reference's checkpoint key names, so the same
weights drive the oracle, the reference (in the build container) and the HIP modules.
Test infrastructure only -- see oracle/__init__.py.

Follows (file:line in /root/reference):
  UNet3DConditionModel.forward            animatediff/models/unet.py:458-621
  CrossAttnDownBlock3D / DownBlock3D      animatediff/models/unet_blocks.py:384-423, 495-523
  UNetMidBlock3DCrossAttn                 animatediff/models/unet_blocks.py:273-280
  CrossAttnUpBlock3D / UpBlock3D          animatediff/models/unet_blocks.py:623-669, 737-762
  ResnetBlock3D / Up/Downsample3D         animatediff/models/resnet.py:188-218, 49-82, 100-108
  Transformer3DModel / BasicTransformerBlock  animatediff/models/attention.py:120-167, 254-300
  VanillaTemporalModule .. VersatileAttention animatediff/models/motion_module.py:79-84,136-160,212-224,272-329
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import nn_ops as ops


@dataclass
class UNet3DConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    cross_attention_dim: int = 768
    attention_heads: int = 8            # SD1.5 config attention_head_dim=8 is used as the head COUNT
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    time_cond_proj_dim: Optional[int] = None   # 256 for the native-LCM UNet
    # animatediff additions (configs/inference/inference-v{1,2}.yaml)
    use_inflated_groupnorm: bool = False       # v2: True (per-frame GN); v1: statistics over frames
    motion_module_mid_block: bool = False      # v2: True
    motion_heads: int = 8
    motion_pe_max_len: int = 24                # v1 24, v2 32
    down_has_attn: Tuple[bool, ...] = (True, True, True, False)

    @staticmethod
    def v1(**kw) -> "UNet3DConfig":
        return UNet3DConfig(**kw)

    @staticmethod
    def v2(**kw) -> "UNet3DConfig":
        base = dict(use_inflated_groupnorm=True, motion_module_mid_block=True, motion_pe_max_len=32)
        base.update(kw)
        return UNet3DConfig(**base)

    @property
    def up_has_attn(self) -> Tuple[bool, ...]:
        return tuple(reversed(self.down_has_attn))


# ------------------------------------------------------------------------------- weights
def motion_pe_table(d_model: int, max_len: int) -> torch.Tensor:
    """PositionalEncoding buffer (motion_module.py:235-241): [1, max_len, d_model]."""
    position = torch.arange(max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(1, max_len, d_model)
    pe[0, :, 0::2] = torch.sin(position * div_term)
    pe[0, :, 1::2] = torch.cos(position * div_term)
    return pe


def unet3d_param_shapes(cfg: UNet3DConfig, include_dead: bool = False) -> Dict[str, Tuple[int, ...]]:
    """Checkpoint keys -> shapes, in the reference's naming. `include_dead` adds the
    unused to_q/to_k/to_v/to_out of BasicTransformerBlock-as-Attention."""
    sh: Dict[str, Tuple[int, ...]] = {}
    boc = cfg.block_out_channels
    temb = boc[0] * 4

    def conv(p, cin, cout, k=3):
        sh[p + ".weight"] = (cout, cin, k, k)
        sh[p + ".bias"] = (cout,)

    def lin(p, cin, cout, bias=True):
        sh[p + ".weight"] = (cout, cin)
        if bias:
            sh[p + ".bias"] = (cout,)

    def norm(p, c):
        sh[p + ".weight"] = (c,)
        sh[p + ".bias"] = (c,)

    def resnet(p, cin, cout):
        norm(p + ".norm1", cin)
        conv(p + ".conv1", cin, cout)
        lin(p + ".time_emb_proj", temb, cout)
        norm(p + ".norm2", cout)
        conv(p + ".conv2", cout, cout)
        if cin != cout:
            conv(p + ".conv_shortcut", cin, cout, 1)

    def attn(p, c, ctx=None):
        lin(p + ".to_q", c, c, False)
        lin(p + ".to_k", ctx or c, c, False)
        lin(p + ".to_v", ctx or c, c, False)
        lin(p + ".to_out.0", c, c)

    def ff(p, c):
        lin(p + ".net.0.proj", c, 8 * c)
        lin(p + ".net.2", 4 * c, c)

    def transformer(p, c):
        norm(p + ".norm", c)
        conv(p + ".proj_in", c, c, 1)
        b = p + ".transformer_blocks.0"
        if include_dead:
            # Attention(query_dim=c) defaults: heads=8, dim_head=64 -> inner 512
            lin(b + ".to_q", c, 512, False)
            lin(b + ".to_k", c, 512, False)
            lin(b + ".to_v", c, 512, False)
            lin(b + ".to_out.0", 512, c)
        attn(b + ".attn1", c)
        norm(b + ".norm1", c)
        attn(b + ".attn2", c, cfg.cross_attention_dim)
        norm(b + ".norm2", c)
        ff(b + ".ff", c)
        norm(b + ".norm3", c)
        conv(p + ".proj_out", c, c, 1)

    def motion(p, c):
        t = p + ".temporal_transformer"
        norm(t + ".norm", c)
        lin(t + ".proj_in", c, c)
        b = t + ".transformer_blocks.0"
        for i in range(2):
            attn(b + f".attention_blocks.{i}", c)
            sh[b + f".attention_blocks.{i}.pos_encoder.pe"] = (1, cfg.motion_pe_max_len, c)
            norm(b + f".norms.{i}", c)
        ff(b + ".ff", c)
        norm(b + ".ff_norm", c)
        lin(t + ".proj_out", c, c)

    conv("conv_in", cfg.in_channels, boc[0])
    lin("time_embedding.linear_1", boc[0], temb)
    lin("time_embedding.linear_2", temb, temb)
    if cfg.time_cond_proj_dim:
        lin("time_embedding.cond_proj", cfg.time_cond_proj_dim, boc[0], False)
    ch = boc[0]
    for i, co in enumerate(boc):
        for j in range(cfg.layers_per_block):
            resnet(f"down_blocks.{i}.resnets.{j}", ch if j == 0 else co, co)
            if cfg.down_has_attn[i]:
                transformer(f"down_blocks.{i}.attentions.{j}", co)
            motion(f"down_blocks.{i}.motion_modules.{j}", co)
        if i < len(boc) - 1:
            conv(f"down_blocks.{i}.downsamplers.0.conv", co, co)
        ch = co
    resnet("mid_block.resnets.0", boc[-1], boc[-1])
    transformer("mid_block.attentions.0", boc[-1])
    if cfg.motion_module_mid_block:
        motion("mid_block.motion_modules.0", boc[-1])
    resnet("mid_block.resnets.1", boc[-1], boc[-1])
    rev = list(reversed(boc))
    prev = rev[0]
    for i, co in enumerate(rev):
        cin_skip = rev[min(i + 1, len(boc) - 1)]
        for j in range(cfg.layers_per_block + 1):
            skip = cin_skip if j == cfg.layers_per_block else co
            rin = prev if j == 0 else co
            resnet(f"up_blocks.{i}.resnets.{j}", rin + skip, co)
            if cfg.up_has_attn[i]:
                transformer(f"up_blocks.{i}.attentions.{j}", co)
            motion(f"up_blocks.{i}.motion_modules.{j}", co)
        if i < len(boc) - 1:
            conv(f"up_blocks.{i}.upsamplers.0.conv", co, co)
        prev = co
    norm("conv_norm_out", boc[0])
    conv("conv_out", boc[0], cfg.out_channels)
    return sh


def init_from_shapes(shapes: Dict[str, Tuple[int, ...]], seed: int = 0, std_scale: float = 1.0) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights for a {key: shape} table (no checkpoints exist offline):
    N(0, 1/fan_in) matrices/convs, norm weights 1 + 0.1 N, biases 0.02 N, sinusoidal `pe` buffers.
    Motion-module proj_out and ControlNet zero-convs come out NON-zero (real checkpoints
    are non-zero there; zeros would hide bugs). Deterministic for a given torch build."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for k, s in shapes.items():
        if k.endswith("pos_encoder.pe"):
            sd[k] = motion_pe_table(s[2], s[1])
        elif len(s) == 1:
            r = torch.randn(s, generator=g)
            sd[k] = 1.0 + 0.1 * r if k.endswith(".weight") else 0.02 * r
        else:
            fan_in = 1
            for d in s[1:]:
                fan_in *= d
            sd[k] = torch.randn(s, generator=g) * (std_scale / math.sqrt(fan_in))
    return sd


def init_unet3d_weights(cfg: UNet3DConfig, seed: int = 0, std_scale: float = 1.0) -> Dict[str, torch.Tensor]:
    return init_from_shapes(unet3d_param_shapes(cfg), seed, std_scale)


def sub_state_dict(sd: Dict[str, torch.Tensor], prefix: str) -> Dict[str, torch.Tensor]:
    """Entries under `prefix.` with the prefix removed (module-level fixtures)."""
    n = len(prefix) + 1
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix + ".")}


# ------------------------------------------------------------------------------- forward
def _to4(x5):
    b, c, f, h, w = x5.shape
    return x5.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)


def _to5(x4, f):
    bf, c, h, w = x4.shape
    return x4.reshape(bf // f, f, c, h, w).permute(0, 2, 1, 3, 4)


def _conv5(sd, p, x5, stride=1):
    return _to5(ops.conv2d(sd, p, _to4(x5), stride=stride), x5.shape[2])


def _gn5(sd, p, x5, cfg: UNet3DConfig, eps):
    if cfg.use_inflated_groupnorm:
        return _to5(ops.group_norm(sd, p, _to4(x5), cfg.norm_num_groups, eps), x5.shape[2])
    return ops.group_norm(sd, p, x5, cfg.norm_num_groups, eps)  # statistics over (c/g, f, h, w)


def resnet_block(sd, p, x5, temb, cfg: UNet3DConfig):
    h = F.silu(_gn5(sd, p + ".norm1", x5, cfg, cfg.norm_eps))
    h = _conv5(sd, p + ".conv1", h)
    h = h + ops.linear(sd, p + ".time_emb_proj", F.silu(temb))[:, :, None, None, None]
    h = F.silu(_gn5(sd, p + ".norm2", h, cfg, cfg.norm_eps))
    h = _conv5(sd, p + ".conv2", h)
    if p + ".conv_shortcut.weight" in sd:
        x5 = _conv5(sd, p + ".conv_shortcut", x5)
    return x5 + h  # output_scale_factor = 1.0 everywhere in SD1.5


def transformer_block(sd, p, x5, ehs, cfg: UNet3DConfig, ip=None):
    """Transformer3DModel (attention.py:120-167) with one BasicTransformerBlock (:254-300)."""
    b, c, f, h, w = x5.shape
    x4 = _to4(x5)
    ctx = ehs.repeat_interleave(f, dim=0)  # 'b n c -> (b f) n c'
    res = x4
    y = ops.group_norm(sd, p + ".norm", x4, cfg.norm_num_groups, 1e-6)
    y = ops.conv2d(sd, p + ".proj_in", y)
    y = y.permute(0, 2, 3, 1).reshape(b * f, h * w, c)
    t = p + ".transformer_blocks.0"
    y = y + ops.attention(sd, t + ".attn1", ops.layer_norm(sd, t + ".norm1", y), None, cfg.attention_heads)
    ipw = ip.get(t + ".attn2") if ip else None
    y = y + ops.attention(sd, t + ".attn2", ops.layer_norm(sd, t + ".norm2", y), ctx, cfg.attention_heads, ip=ipw)
    y = y + ops.feed_forward(sd, t + ".ff", ops.layer_norm(sd, t + ".norm3", y))
    y = y.reshape(b * f, h, w, c).permute(0, 3, 1, 2)
    y = ops.conv2d(sd, p + ".proj_out", y)
    return _to5(y + res, f)


def motion_module(sd, p, x5, cfg: UNet3DConfig):
    """VanillaTemporalModule -> TemporalTransformer3DModel (motion_module.py:136-160)."""
    b, c, f, h, w = x5.shape
    t = p + ".temporal_transformer"
    x4 = _to4(x5)
    res = x4
    y = ops.group_norm(sd, t + ".norm", x4, cfg.norm_num_groups, 1e-6)
    y = y.permute(0, 2, 3, 1).reshape(b * f, h * w, c)
    y = ops.linear(sd, t + ".proj_in", y)
    blk = t + ".transformer_blocks.0"
    d = h * w
    for i in range(2):
        ab = blk + f".attention_blocks.{i}"
        n = ops.layer_norm(sd, blk + f".norms.{i}", y)
        # '(b f) d c -> (b d) f c', + pe[:, :f], temporal self-attention, back (:285-327)
        z = n.reshape(b, f, d, c).permute(0, 2, 1, 3).reshape(b * d, f, c)
        z = z + sd[ab + ".pos_encoder.pe"][:, :f]
        z = ops.attention(sd, ab, z, None, cfg.motion_heads)
        z = z.reshape(b, d, f, c).permute(0, 2, 1, 3).reshape(b * f, d, c)
        y = y + z
    y = y + ops.feed_forward(sd, blk + ".ff", ops.layer_norm(sd, blk + ".ff_norm", y))
    y = ops.linear(sd, t + ".proj_out", y)
    y = y.reshape(b * f, h, w, c).permute(0, 3, 1, 2)
    return _to5(y + res, f)


def unet3d_forward(sd, cfg: UNet3DConfig, sample: torch.Tensor, timestep, encoder_hidden_states: torch.Tensor,
                   down_block_additional_residuals: Optional[Sequence[torch.Tensor]] = None,
                   mid_block_additional_residual: Optional[torch.Tensor] = None,
                   timestep_cond: Optional[torch.Tensor] = None, ip: Optional[dict] = None) -> torch.Tensor:
    """sample [b,4,f,h,w], timestep scalar or [b], encoder_hidden_states [b,L,768] -> eps [b,4,f,h,w].

    ip: {"<attn2 module path>": {"to_k_ip","to_v_ip","scale","num_tokens"}} for IP-Adapter sites.
    ControlNet residuals broadcast over b when given with b=1 (unet.py:573)."""
    sample = sample.float()
    ehs = encoder_hidden_states.float()
    b = sample.shape[0]
    boc = cfg.block_out_channels
    t = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1).expand(b) if not torch.is_tensor(timestep) \
        else timestep.float().reshape(-1).expand(b)
    emb = ops.time_embedding(sd, "time_embedding", ops.timestep_sinusoid(t, boc[0]),
                             None if timestep_cond is None else timestep_cond.float())
    x = _conv5(sd, "conv_in", sample)
    skips = [x]
    for i in range(len(boc)):
        for j in range(cfg.layers_per_block):
            x = resnet_block(sd, f"down_blocks.{i}.resnets.{j}", x, emb, cfg)
            if cfg.down_has_attn[i]:
                x = transformer_block(sd, f"down_blocks.{i}.attentions.{j}", x, ehs, cfg, ip)
            x = motion_module(sd, f"down_blocks.{i}.motion_modules.{j}", x, cfg)
            skips.append(x)
        if i < len(boc) - 1:
            x = _conv5(sd, f"down_blocks.{i}.downsamplers.0.conv", x, stride=2)
            skips.append(x)
    if down_block_additional_residuals is not None:
        skips = [s + r.float() for s, r in zip(skips, down_block_additional_residuals)]
    x = resnet_block(sd, "mid_block.resnets.0", x, emb, cfg)
    x = transformer_block(sd, "mid_block.attentions.0", x, ehs, cfg, ip)
    if cfg.motion_module_mid_block:
        x = motion_module(sd, "mid_block.motion_modules.0", x, cfg)
    x = resnet_block(sd, "mid_block.resnets.1", x, emb, cfg)
    if mid_block_additional_residual is not None:
        x = x + mid_block_additional_residual.float()
    for i in range(len(boc)):
        for j in range(cfg.layers_per_block + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet_block(sd, f"up_blocks.{i}.resnets.{j}", x, emb, cfg)
            if cfg.up_has_attn[i]:
                x = transformer_block(sd, f"up_blocks.{i}.attentions.{j}", x, ehs, cfg, ip)
            x = motion_module(sd, f"up_blocks.{i}.motion_modules.{j}", x, cfg)
        if i < len(boc) - 1:
            x = x.repeat_interleave(2, dim=3).repeat_interleave(2, dim=4)  # nearest x2 on (h, w)
            x = _conv5(sd, f"up_blocks.{i}.upsamplers.0.conv", x)
    x = F.silu(_gn5(sd, "conv_norm_out", x, cfg, cfg.norm_eps))
    return _conv5(sd, "conv_out", x)
