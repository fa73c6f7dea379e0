"""fp32 oracle of the CLIP encoders the reference drives (transformers CLIPTextModel and
CLIPVisionModelWithProjection).

Test infrastructure only.  transformers is a THIRD-PARTY dependency of the reference (env.yml); it IS
importable in the build container, so this restatement is PINNED: tests/golden/make_clip_golden.py runs the
real transformers modules on tiny seeded configurations and tests/test_oracle_golden.py-style checks
(tests/test_clip_cpu.py) compare this file against those fixtures (tests/golden/clip_*.npz).
Call sites in the reference:
  modules/controlanimate_pipeline.py:133-135   Compel(tokenizer, text_encoder)(prompt) -> prompt embeds [1,77,768]
  modules/ip_adapter.py:72-75, 187-198         CLIPVisionModelWithProjection(pixel_values).image_embeds [B,1024]
Architecture (transformers/models/clip/modeling_clip.py): pre-LN transformer; text: token + position
embeddings, causal mask, final LayerNorm, pooled = state at the EOS token; vision: 14x14 patch conv (no
bias), class token, position embeddings, `pre_layrnorm` (sic), post LayerNorm on the class token, linear
projection (no bias).  MLP activation quick_gelu (ViT-L) or gelu (ViT-H).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class CLIPTextConfig:
    vocab_size: int = 49408
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    max_position_embeddings: int = 77
    hidden_act: str = "quick_gelu"
    layer_norm_eps: float = 1e-5
    eos_token_id: int = 2  # the SD1.5 text encoder config keeps the legacy value: pooled = state at argmax(input_ids)


@dataclass
class CLIPVisionConfig:
    hidden_size: int = 1280
    intermediate_size: int = 5120
    num_hidden_layers: int = 32
    num_attention_heads: int = 16
    image_size: int = 224
    patch_size: int = 14
    projection_dim: int = 1024
    hidden_act: str = "gelu"
    layer_norm_eps: float = 1e-5


def _act(x, name):
    if name == "quick_gelu":
        return x * torch.sigmoid(1.702 * x)
    if name == "gelu":
        return F.gelu(x)
    raise ValueError(name)


def _ln(sd, p, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _encoder(sd, prefix, x, layers, heads, act, eps, causal, collect, attention_mask=None):
    b, n, c = x.shape
    d = c // heads
    mask = torch.full((n, n), float("-inf")).triu(1) if causal else None
    if attention_mask is not None:  # [B, L], 1 = attend: additive [B,1,1,L] on the KEYS (transformers CLIPTextModel)
        km = torch.zeros(b, 1, 1, n).masked_fill(~attention_mask.bool()[:, None, None, :], float("-inf"))
        mask = km if mask is None else mask + km
    for i in range(layers):
        p = f"{prefix}.layers.{i}"
        h = _ln(sd, p + ".layer_norm1", x, eps)
        q, k, v = (_lin(sd, f"{p}.self_attn.{w}_proj", h).view(b, n, heads, d).transpose(1, 2) for w in "qkv")
        s = (q @ k.transpose(-1, -2)) * d ** -0.5
        if mask is not None:
            s = s + mask
        o = (s.softmax(-1) @ v).transpose(1, 2).reshape(b, n, c)
        x = x + _lin(sd, p + ".self_attn.out_proj", o)
        h = _ln(sd, p + ".layer_norm2", x, eps)
        x = x + _lin(sd, p + ".mlp.fc2", _act(_lin(sd, p + ".mlp.fc1", h), act))
        collect.append(x)
    return x


def clip_text_forward(sd: Dict[str, torch.Tensor], cfg: CLIPTextConfig, input_ids: torch.Tensor, attention_mask=None):
    """-> (last_hidden_state [B,L,C], pooler_output [B,C], hidden_states tuple of L+1 tensors)."""
    b, n = input_ids.shape
    x = sd["text_model.embeddings.token_embedding.weight"][input_ids] + sd["text_model.embeddings.position_embedding.weight"][:n]
    hs = [x]
    x = _encoder(sd, "text_model.encoder", x, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.hidden_act, cfg.layer_norm_eps, True, hs,
                 attention_mask=attention_mask)
    last = _ln(sd, "text_model.final_layer_norm", x, cfg.layer_norm_eps)
    if cfg.eos_token_id == 2:
        pos = input_ids.argmax(-1)
    else:
        pos = (input_ids == cfg.eos_token_id).int().argmax(-1)
    return last, last[torch.arange(b), pos], tuple(hs)


def clip_vision_forward(sd: Dict[str, torch.Tensor], cfg: CLIPVisionConfig, pixel_values: torch.Tensor):
    """-> (image_embeds [B,proj], last_hidden_state [B,N+1,C], pooled [B,C])."""
    b = pixel_values.shape[0]
    pe = F.conv2d(pixel_values, sd["vision_model.embeddings.patch_embedding.weight"], stride=cfg.patch_size)
    pe = pe.flatten(2).transpose(1, 2)
    cls = sd["vision_model.embeddings.class_embedding"].expand(b, 1, -1)
    x = torch.cat([cls, pe], 1) + sd["vision_model.embeddings.position_embedding.weight"]
    x = _ln(sd, "vision_model.pre_layrnorm", x, cfg.layer_norm_eps)
    x = _encoder(sd, "vision_model.encoder", x, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.hidden_act, cfg.layer_norm_eps, False, [])
    pooled = _ln(sd, "vision_model.post_layernorm", x[:, 0], cfg.layer_norm_eps)
    return F.linear(pooled, sd["visual_projection.weight"]), x, pooled


def clip_param_shapes(cfg, kind: str) -> Dict[str, Tuple[int, ...]]:
    sh: Dict[str, Tuple[int, ...]] = {}
    c, inter = cfg.hidden_size, cfg.intermediate_size
    pre = "text_model" if kind == "text" else "vision_model"
    for i in range(cfg.num_hidden_layers):
        p = f"{pre}.encoder.layers.{i}"
        for w in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sh[f"{p}.self_attn.{w}.weight"], sh[f"{p}.self_attn.{w}.bias"] = (c, c), (c,)
        for ln in ("layer_norm1", "layer_norm2"):
            sh[f"{p}.{ln}.weight"], sh[f"{p}.{ln}.bias"] = (c,), (c,)
        sh[f"{p}.mlp.fc1.weight"], sh[f"{p}.mlp.fc1.bias"] = (inter, c), (inter,)
        sh[f"{p}.mlp.fc2.weight"], sh[f"{p}.mlp.fc2.bias"] = (c, inter), (c,)
    if kind == "text":
        sh["text_model.embeddings.token_embedding.weight"] = (cfg.vocab_size, c)
        sh["text_model.embeddings.position_embedding.weight"] = (cfg.max_position_embeddings, c)
        sh["text_model.final_layer_norm.weight"], sh["text_model.final_layer_norm.bias"] = (c,), (c,)
    else:
        n = (cfg.image_size // cfg.patch_size) ** 2 + 1
        sh["vision_model.embeddings.class_embedding"] = (c,)
        sh["vision_model.embeddings.patch_embedding.weight"] = (c, 3, cfg.patch_size, cfg.patch_size)
        sh["vision_model.embeddings.position_embedding.weight"] = (n, c)
        for ln in ("pre_layrnorm", "post_layernorm"):
            sh[f"vision_model.{ln}.weight"], sh[f"vision_model.{ln}.bias"] = (c,), (c,)
        sh["visual_projection.weight"] = (cfg.projection_dim, c)
    return sh


def init_clip_weights(cfg, kind: str, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shape in clip_param_shapes(cfg, kind).items():
        if "layer_norm" in k or "layrnorm" in k or "layernorm" in k:
            sd[k] = (1.0 + 0.1 * torch.randn(shape, generator=g)) if k.endswith("weight") else 0.1 * torch.randn(shape, generator=g)
        elif k.endswith(".bias"):
            sd[k] = 0.02 * torch.randn(shape, generator=g)
        elif "embedding" in k:
            sd[k] = 0.3 * torch.randn(shape, generator=g) if "patch" not in k else torch.randn(shape, generator=g) * (3 * shape[-1] ** 2) ** -0.5
        else:
            sd[k] = torch.randn(shape, generator=g) * shape[-1] ** -0.5
    return sd
