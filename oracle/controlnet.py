"""fp32 oracle of diffusers==0.23.0 ControlNetModel / MultiControlNetModel (SD1.5 configuration).

Test infrastructure only.  diffusers is a THIRD-PARTY dependency that is absent from
/root/reference (pinned at env.yml:120) and not installable here: this file restates the published
algorithm.  PINNING: the encoder + mid block are the SD1.5 UNet encoder, whose arithmetic the
reference carries itself (animatediff/models/unet_blocks.py:173-523 with use_motion_module=False, one frame per image):
tests/golden/make_controlnet_golden.py assembles a ControlNet from those reference blocks and
tests/test_oracle_golden.py::test_controlnet_oracle_matches_reference_blocks checks this file against it (reduced and
full SD1.5 width, plain / guess-mode scaling, CNAttnProcessor2_0 token strip; <= 3e-5).  Still restatement-only: the 21
plain convolutions around the blocks (conv_in, hint embedding, 13 zero-convs), the logspace guess-mode scales and the sum
over nets of MultiControlNetModel.  What the reference owns besides the blocks and this file follows:
  modules/controlresiduals_pipeline.py:278-316   rearranges, prompt repetition quirk, (b f) batching
  animatediff/utils/convert_from_ckpt.py:514-554 checkpoint key names (controlnet_cond_embedding.*,
                                                 controlnet_down_blocks.0-11, controlnet_mid_block)
  modules/attention_processor.py:571-646         CNAttnProcessor2_0 (drops the 4 IP tokens)
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import nn_ops as ops


@dataclass
class ControlNetConfig:
    in_channels: int = 4
    conditioning_channels: int = 3
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    cond_embedding_channels: Tuple[int, ...] = (16, 32, 96, 256)
    layers_per_block: int = 2
    cross_attention_dim: int = 768
    attention_heads: int = 8
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    down_has_attn: Tuple[bool, ...] = (True, True, True, False)


def controlnet_param_shapes(cfg: ControlNetConfig) -> Dict[str, Tuple[int, ...]]:
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

    def transformer(p, c):
        norm(p + ".norm", c)
        conv(p + ".proj_in", c, c, 1)
        b = p + ".transformer_blocks.0"
        for name, ctx in (("attn1", c), ("attn2", cfg.cross_attention_dim)):
            lin(b + f".{name}.to_q", c, c, False)
            lin(b + f".{name}.to_k", ctx, c, False)
            lin(b + f".{name}.to_v", ctx, c, False)
            lin(b + f".{name}.to_out.0", c, c)
        for i in (1, 2, 3):
            norm(b + f".norm{i}", c)
        lin(b + ".ff.net.0.proj", c, 8 * c)
        lin(b + ".ff.net.2", 4 * c, c)
        conv(p + ".proj_out", c, c, 1)

    conv("conv_in", cfg.in_channels, boc[0])
    lin("time_embedding.linear_1", boc[0], temb)
    lin("time_embedding.linear_2", temb, temb)
    ce = cfg.cond_embedding_channels
    conv("controlnet_cond_embedding.conv_in", cfg.conditioning_channels, ce[0])
    bi = 0
    for i in range(len(ce) - 1):
        conv(f"controlnet_cond_embedding.blocks.{bi}", ce[i], ce[i])
        conv(f"controlnet_cond_embedding.blocks.{bi + 1}", ce[i], ce[i + 1])  # stride 2
        bi += 2
    conv("controlnet_cond_embedding.conv_out", ce[-1], boc[0])
    zi = 0
    conv(f"controlnet_down_blocks.{zi}", boc[0], boc[0], 1)
    zi += 1
    ch = boc[0]
    for i, co in enumerate(boc):
        for j in range(cfg.layers_per_block):
            resnet(f"down_blocks.{i}.resnets.{j}", ch if j == 0 else co, co)
            if cfg.down_has_attn[i]:
                transformer(f"down_blocks.{i}.attentions.{j}", co)
            conv(f"controlnet_down_blocks.{zi}", co, co, 1)
            zi += 1
        if i < len(boc) - 1:
            conv(f"down_blocks.{i}.downsamplers.0.conv", co, co)
            conv(f"controlnet_down_blocks.{zi}", co, co, 1)
            zi += 1
        ch = co
    resnet("mid_block.resnets.0", boc[-1], boc[-1])
    transformer("mid_block.attentions.0", boc[-1])
    resnet("mid_block.resnets.1", boc[-1], boc[-1])
    conv("controlnet_mid_block", boc[-1], boc[-1], 1)
    return sh


def init_controlnet_weights(cfg: ControlNetConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights; zero-convs are non-zero (trained checkpoints are)."""
    from .unet3d import init_from_shapes
    return init_from_shapes(controlnet_param_shapes(cfg), seed)


def _resnet2d(sd, p, x, temb, cfg):
    h = F.silu(ops.group_norm(sd, p + ".norm1", x, cfg.norm_num_groups, cfg.norm_eps))
    h = ops.conv2d(sd, p + ".conv1", h)
    h = h + ops.linear(sd, p + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    h = F.silu(ops.group_norm(sd, p + ".norm2", h, cfg.norm_num_groups, cfg.norm_eps))
    h = ops.conv2d(sd, p + ".conv2", h)
    if p + ".conv_shortcut.weight" in sd:
        x = ops.conv2d(sd, p + ".conv_shortcut", x)
    return x + h


def _transformer2d(sd, p, x, ehs, cfg, strip_tokens):
    B, c, h, w = x.shape
    res = x
    y = ops.group_norm(sd, p + ".norm", x, cfg.norm_num_groups, 1e-6)
    y = ops.conv2d(sd, p + ".proj_in", y)
    y = y.permute(0, 2, 3, 1).reshape(B, h * w, c)
    t = p + ".transformer_blocks.0"
    y = y + ops.attention(sd, t + ".attn1", ops.layer_norm(sd, t + ".norm1", y), None, cfg.attention_heads)
    y = y + ops.attention(sd, t + ".attn2", ops.layer_norm(sd, t + ".norm2", y), ehs, cfg.attention_heads,
                          strip_tokens=strip_tokens)
    y = y + ops.feed_forward(sd, t + ".ff", ops.layer_norm(sd, t + ".norm3", y))
    y = y.reshape(B, h, w, c).permute(0, 3, 1, 2)
    return ops.conv2d(sd, p + ".proj_out", y) + res


def controlnet_cond_embedding(sd, cfg: ControlNetConfig, cond: torch.Tensor) -> torch.Tensor:
    """ControlNetConditioningEmbedding: conv_in -> SiLU -> [conv, SiLU, conv s2, SiLU]x3 -> conv_out."""
    p = "controlnet_cond_embedding"
    e = F.silu(ops.conv2d(sd, p + ".conv_in", cond))
    for i in range(2 * (len(cfg.cond_embedding_channels) - 1)):
        e = F.silu(ops.conv2d(sd, p + f".blocks.{i}", e, stride=2 if i % 2 else 1))
    return ops.conv2d(sd, p + ".conv_out", e)


def controlnet_forward(sd, cfg: ControlNetConfig, sample: torch.Tensor, timestep, encoder_hidden_states: torch.Tensor,
                       controlnet_cond: torch.Tensor, conditioning_scale: float = 1.0, guess_mode: bool = False,
                       strip_tokens: int = 0) -> Tuple[List[torch.Tensor], torch.Tensor]:
    """sample [B,4,h,w], ehs [B,L,768], cond [B,3,8h,8w] in [0,1] -> (12 down residuals, mid residual)."""
    sample = sample.float()
    ehs = encoder_hidden_states.float()
    B = sample.shape[0]
    boc = cfg.block_out_channels
    t = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1).expand(B)
    emb = ops.time_embedding(sd, "time_embedding", ops.timestep_sinusoid(t, boc[0]), None)
    x = ops.conv2d(sd, "conv_in", sample) + controlnet_cond_embedding(sd, cfg, controlnet_cond.float())
    outs = [x]
    for i in range(len(boc)):
        for j in range(cfg.layers_per_block):
            x = _resnet2d(sd, f"down_blocks.{i}.resnets.{j}", x, emb, cfg)
            if cfg.down_has_attn[i]:
                x = _transformer2d(sd, f"down_blocks.{i}.attentions.{j}", x, ehs, cfg, strip_tokens)
            outs.append(x)
        if i < len(boc) - 1:
            x = ops.conv2d(sd, f"down_blocks.{i}.downsamplers.0.conv", x, stride=2)
            outs.append(x)
    x = _resnet2d(sd, "mid_block.resnets.0", x, emb, cfg)
    x = _transformer2d(sd, "mid_block.attentions.0", x, ehs, cfg, strip_tokens)
    x = _resnet2d(sd, "mid_block.resnets.1", x, emb, cfg)
    down = [ops.conv2d(sd, f"controlnet_down_blocks.{i}", o) for i, o in enumerate(outs)]
    mid = ops.conv2d(sd, "controlnet_mid_block", x)
    if guess_mode:
        scales = torch.logspace(-1, 0, len(down) + 1) * conditioning_scale
        down = [d * s for d, s in zip(down, scales)]
        mid = mid * scales[-1]
    else:
        down = [d * conditioning_scale for d in down]
        mid = mid * conditioning_scale
    return down, mid


def multi_controlnet_residuals(nets: Sequence[dict], cfg: ControlNetConfig, control_model_input: torch.Tensor, t,
                               controlnet_prompt_embeds: torch.Tensor, frame_count: int,
                               prep_images: Sequence[torch.Tensor], cond_scale: Sequence[float],
                               guess_mode: bool, strip_tokens: int = 0):
    """MultiControlNetResidualsPipeline.__call__ (modules/controlresiduals_pipeline.py:278-316).

    control_model_input [b,4,f,h,w]; prompt embeds [b,L,768]; prep_images: per net [(b f),3,H,W].
    NOTE the reference's prompt-order quirk: embeds are tiled with
    torch.cat([embeds]*frame_count) -> [e0,e1,e0,e1,...] while images are (b f)-ordered."""
    b, c, f, h, w = control_model_input.shape
    x = control_model_input.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    embeds = torch.cat([controlnet_prompt_embeds] * frame_count)
    down_sum, mid_sum = None, None
    for sd, img, scale in zip(nets, prep_images, cond_scale):
        d, m = controlnet_forward(sd, cfg, x, t, embeds, img, scale, guess_mode, strip_tokens)
        if down_sum is None:
            down_sum, mid_sum = d, m
        else:
            down_sum = [a + bb for a, bb in zip(down_sum, d)]
            mid_sum = mid_sum + m

    def to5(tn):
        bf, cc, hh, ww = tn.shape
        return tn.reshape(bf // frame_count, frame_count, cc, hh, ww).permute(0, 2, 1, 3, 4)

    return [to5(d) for d in down_sum], to5(mid_sum)
