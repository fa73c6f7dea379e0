"""fp32 oracle of diffusers==0.23.0 AutoencoderKL (SD1.5 VAE: encode -> DiagonalGaussian, decode).

Test infrastructure only (imported by tests/, never by the product).  diffusers is a THIRD-PARTY
dependency absent from /root/reference (pinned at env.yml:120) and not installable here: this file
restates the published architecture -- PARITY UNPINNED for the VAE arithmetic itself.  What the
reference owns and this file follows:
  animatediff/pipelines/controlanimation_pipeline.py:501-514  decode_latents: 1/scaling_factor, one
        frame at a time, (x/2 + 0.5).clamp(0, 1)
  animatediff/pipelines/controlanimation_pipeline.py:566-588  encode: latent_dist.sample(generator) *
        scaling_factor, one frame at a time, the SAME generator threaded through all frames
  animatediff/utils/convert_from_ckpt.py:560-662              checkpoint key names (encoder/decoder
        .conv_in, .down_blocks.i.resnets.j, .downsamplers.0.conv, .mid_block.{resnets,attentions.0},
        .up_blocks.i.{resnets.j,upsamplers.0.conv}, .conv_norm_out, .conv_out, quant_conv,
        post_quant_conv; attention weights under the deprecated names query/key/value/proj_attn,
        which diffusers 0.23 renames to to_q/to_k/to_v/to_out.0 on load)

Architecture restated (diffusers/models/{autoencoder_kl,vae,unet_2d_blocks,resnet,attention_processor}.py
at tag v0.23.0):
  Encoder: conv_in 3->128; 4 DownEncoderBlock2D (2 ResnetBlock2D each, eps 1e-6, no time embedding;
           blocks 0-2 end with Downsample2D(padding=0): F.pad(x,(0,1,0,1)) + conv3x3 stride 2);
           UNetMidBlock2D (resnet, single-head attention over h*w with GroupNorm + residual, resnet);
           GroupNorm(32, eps 1e-6) + SiLU + conv_out -> 2*latent; quant_conv 1x1.
  Decoder: post_quant_conv 1x1; conv_in 4->512; mid block; 4 UpDecoderBlock2D (3 resnets each, blocks
           0-2 end with nearest x2 + conv3x3); GroupNorm + SiLU + conv_out -> 3.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from . import nn_ops as ops


@dataclass
class VAEConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215
    eps: float = 1e-6


def vae_param_shapes(cfg: VAEConfig) -> Dict[str, Tuple[int, ...]]:
    sh: Dict[str, Tuple[int, ...]] = {}
    boc = cfg.block_out_channels

    def conv(p, cin, cout, k=3):
        sh[p + ".weight"] = (cout, cin, k, k)
        sh[p + ".bias"] = (cout,)

    def lin(p, cin, cout):
        sh[p + ".weight"] = (cout, cin)
        sh[p + ".bias"] = (cout,)

    def norm(p, c):
        sh[p + ".weight"] = (c,)
        sh[p + ".bias"] = (c,)

    def resnet(p, cin, cout):
        norm(p + ".norm1", cin)
        conv(p + ".conv1", cin, cout)
        norm(p + ".norm2", cout)
        conv(p + ".conv2", cout, cout)
        if cin != cout:
            conv(p + ".conv_shortcut", cin, cout, 1)

    def mid(p, c):
        resnet(p + ".resnets.0", c, c)
        norm(p + ".attentions.0.group_norm", c)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            lin(p + ".attentions.0." + n, c, c)
        resnet(p + ".resnets.1", c, c)

    conv("encoder.conv_in", cfg.in_channels, boc[0])
    cin = boc[0]
    for i, cout in enumerate(boc):
        for j in range(cfg.layers_per_block):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
        if i < len(boc) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", cout, cout)
        cin = cout
    mid("encoder.mid_block", boc[-1])
    norm("encoder.conv_norm_out", boc[-1])
    conv("encoder.conv_out", boc[-1], 2 * cfg.latent_channels)
    conv("quant_conv", 2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
    conv("post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
    rev = tuple(reversed(boc))
    conv("decoder.conv_in", cfg.latent_channels, rev[0])
    mid("decoder.mid_block", rev[0])
    cin = rev[0]
    for i, cout in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
        if i < len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", cout, cout)
        cin = cout
    norm("decoder.conv_norm_out", rev[-1])
    conv("decoder.conv_out", rev[-1], cfg.out_channels)
    return sh


def init_vae_weights(cfg: VAEConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded random weights at checkpoint-like magnitudes (no checkpoint offline)."""
    from .unet3d import init_from_shapes
    return init_from_shapes(vae_param_shapes(cfg), seed)


def _resnet(sd, p, x, cfg: VAEConfig):
    h = F.silu(ops.group_norm(sd, p + ".norm1", x, cfg.norm_num_groups, cfg.eps))
    h = ops.conv2d(sd, p + ".conv1", h)
    h = F.silu(ops.group_norm(sd, p + ".norm2", h, cfg.norm_num_groups, cfg.eps))
    h = ops.conv2d(sd, p + ".conv2", h)
    if (p + ".conv_shortcut.weight") in sd:
        x = ops.conv2d(sd, p + ".conv_shortcut", x, padding=0)
    return x + h  # output_scale_factor = 1


def _mid_attention(sd, p, x, cfg: VAEConfig):
    """Attention(heads=1, dim_head=C, residual_connection=True, norm_num_groups=32, bias=True)."""
    b, c, hh, ww = x.shape
    hs = ops.group_norm(sd, p + ".group_norm", x.view(b, c, hh * ww), cfg.norm_num_groups, cfg.eps).transpose(1, 2)  # [b, hw, c]
    q, k, v = (ops.linear(sd, p + "." + n, hs) for n in ("to_q", "to_k", "to_v"))
    o = ops.sdpa(q, k, v, heads=1)
    o = ops.linear(sd, p + ".to_out.0", o)
    return o.transpose(1, 2).reshape(b, c, hh, ww) + x  # rescale_output_factor = 1


def _mid(sd, p, x, cfg):
    x = _resnet(sd, p + ".resnets.0", x, cfg)
    x = _mid_attention(sd, p + ".attentions.0", x, cfg)
    return _resnet(sd, p + ".resnets.1", x, cfg)


def vae_encode_moments(sd, cfg: VAEConfig, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """x [B,3,H,W] in [-1,1] -> (mean, logvar) [B,4,H/8,W/8]; logvar clamped to [-30, 20]."""
    h = ops.conv2d(sd, "encoder.conv_in", x)
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        for j in range(cfg.layers_per_block):
            h = _resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}", h, cfg)
        if i < nb - 1:
            h = F.pad(h, (0, 1, 0, 1))
            h = ops.conv2d(sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", h, stride=2, padding=0)
    h = _mid(sd, "encoder.mid_block", h, cfg)
    h = F.silu(ops.group_norm(sd, "encoder.conv_norm_out", h, cfg.norm_num_groups, cfg.eps))
    h = ops.conv2d(sd, "encoder.conv_out", h)
    moments = ops.conv2d(sd, "quant_conv", h, padding=0)
    mean, logvar = moments.chunk(2, dim=1)
    return mean, logvar.clamp(-30.0, 20.0)


def vae_sample(mean: torch.Tensor, logvar: torch.Tensor, generator: Optional[torch.Generator]) -> torch.Tensor:
    """DiagonalGaussianDistribution.sample: mean + exp(0.5 logvar) * randn(mean.shape, generator) (CPU generator
    draws on the CPU, diffusers randn_tensor)."""
    noise = torch.randn(mean.shape, generator=generator, dtype=torch.float32)
    return mean + torch.exp(0.5 * logvar) * noise.to(mean.device)


def vae_decode(sd, cfg: VAEConfig, z: torch.Tensor) -> torch.Tensor:
    """z [B,4,h,w] (already divided by scaling_factor) -> sample [B,3,8h,8w]."""
    h = ops.conv2d(sd, "post_quant_conv", z, padding=0)
    h = ops.conv2d(sd, "decoder.conv_in", h)
    h = _mid(sd, "decoder.mid_block", h, cfg)
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        for j in range(cfg.layers_per_block + 1):
            h = _resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", h, cfg)
        if i < nb - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = ops.conv2d(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", h)
    h = F.silu(ops.group_norm(sd, "decoder.conv_norm_out", h, cfg.norm_num_groups, cfg.eps))
    return ops.conv2d(sd, "decoder.conv_out", h)


def decode_latents(sd, cfg: VAEConfig, latents: torch.Tensor) -> torch.Tensor:
    """controlanimation_pipeline.py:501-514: [b,4,f,h,w] -> video [b,3,f,8h,8w] in [0,1] (fp32)."""
    b, c, f, hh, ww = latents.shape
    z = (latents / cfg.scaling_factor).permute(0, 2, 1, 3, 4).reshape(b * f, c, hh, ww)
    frames = [vae_decode(sd, cfg, z[i:i + 1]) for i in range(b * f)]
    video = torch.cat(frames).view(b, f, -1, 8 * hh, 8 * ww).permute(0, 2, 1, 3, 4)
    return (video / 2 + 0.5).clamp(0, 1)
