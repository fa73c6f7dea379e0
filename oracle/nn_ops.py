"""fp32 building blocks of the oracle (plain torch CPU ops). Test infrastructure only.

Restates diffusers==0.23.0 semantics the reference relies on; the
reference call sites are cited per function.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F


def linear(sd, prefix: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"))


def conv2d(sd, prefix: str, x: torch.Tensor, stride: int = 1, padding: int = 1) -> torch.Tensor:
    w = sd[prefix + ".weight"]
    return F.conv2d(x, w, sd.get(prefix + ".bias"), stride=stride, padding=padding if w.shape[-1] == 3 else 0)


def group_norm(sd, prefix: str, x: torch.Tensor, groups: int, eps: float) -> torch.Tensor:
    """x may be 4-D (per image) or 5-D [b,c,f,h,w] (statistics over f,h,w: plain nn.GroupNorm on
    the video tensor, animatediff/models/resnet.py:150-151)."""
    return F.group_norm(x, groups, sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def layer_norm(sd, prefix: str, x: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], 1e-5)


def sdpa(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int) -> torch.Tensor:
    """[B, Nq, C], [B, Nk, C], [B, Nk, C] -> [B, Nq, C]; softmax(q k^T / sqrt(d)) v per head
    (modules/attention_processor.py:243-257; scale = dim_head**-0.5)."""
    B, nq, c = q.shape
    d = c // heads
    qh = q.reshape(B, nq, heads, d).transpose(1, 2)
    kh = k.reshape(B, -1, heads, d).transpose(1, 2)
    vh = v.reshape(B, -1, heads, d).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) * (d ** -0.5)
    o = torch.softmax(s, dim=-1) @ vh
    return o.transpose(1, 2).reshape(B, nq, c)


def attention(sd, prefix: str, x: torch.Tensor, context: Optional[torch.Tensor], heads: int,
              ip: Optional[dict] = None, strip_tokens: int = 0) -> torch.Tensor:
    """diffusers Attention with the reference's processors.

    context None -> self-attention (AttnProcessor2_0, modules/attention_processor.py:200-272).
    ip = {"to_k_ip": W, "to_v_ip": W, "scale": s, "num_tokens": 4} -> IPAttnProcessor2_0 (:395-492):
      the last num_tokens context rows go through to_k_ip/to_v_ip, out = attn + scale * ip_attn.
    strip_tokens > 0 -> CNAttnProcessor2_0 (:571-646): drop the last tokens of the context.
    """
    q = linear(sd, prefix + ".to_q", x)
    ctx = x if context is None else context
    ip_ctx = None
    if context is not None and ip is not None:
        end = ctx.shape[1] - ip["num_tokens"]
        ctx, ip_ctx = ctx[:, :end], ctx[:, end:]
    if context is not None and strip_tokens:
        ctx = ctx[:, : ctx.shape[1] - strip_tokens]
    k = linear(sd, prefix + ".to_k", ctx)
    v = linear(sd, prefix + ".to_v", ctx)
    o = sdpa(q, k, v, heads)
    if ip_ctx is not None:
        ik = F.linear(ip_ctx, ip["to_k_ip"])
        iv = F.linear(ip_ctx, ip["to_v_ip"])
        o = o + ip["scale"] * sdpa(q, ik, iv, heads)
    return linear(sd, prefix + ".to_out.0", o)


def feed_forward(sd, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """FeedForward(dim, mult=4, 'geglu'): net.0 = GEGLU (proj -> h * gelu(gate), erf GELU),
    net.2 = Linear (animatediff/models/attention.py:303-357)."""
    h, gate = linear(sd, prefix + ".net.0.proj", x).chunk(2, dim=-1)
    return linear(sd, prefix + ".net.2", h * F.gelu(gate))


def timestep_sinusoid(t: torch.Tensor, dim: int) -> torch.Tensor:
    """Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin] (App. A-3)."""
    half = dim // 2
    freq = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    arg = t.float()[:, None] * freq[None]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


def time_embedding(sd, prefix: str, t_emb: torch.Tensor, cond: Optional[torch.Tensor]) -> torch.Tensor:
    """TimestepEmbedding: (+ cond_proj(cond)) -> linear_1 -> SiLU -> linear_2 (App. A-3)."""
    if cond is not None:
        t_emb = t_emb + F.linear(cond, sd[prefix + ".cond_proj.weight"])
    return linear(sd, prefix + ".linear_2", F.silu(linear(sd, prefix + ".linear_1", t_emb)))
