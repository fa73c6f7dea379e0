"""CLIP text and vision encoders on the HIP kernels: the once-per-window
conditioning models the reference takes from transformers.

Mirrors the interfaces the reference uses:
  CLIPTextModel(input_ids)[0] / .last_hidden_state   via Compel, modules/controlanimate_pipeline.py:133-135
  CLIPVisionModelWithProjection(pixel_values).image_embeds      modules/ip_adapter.py:72-75, 187-198
State-dict keys are the transformers 4.x checkpoint names (`text_model.encoder.layers.0.self_attn.q_proj.weight`,
`vision_model.pre_layrnorm.weight`, `visual_projection.weight`); the un-prefixed text keys transformers 5.x
writes are accepted too.  Tokenisation, prompt weighting (Compel) and image preprocessing stay on the host.

Execution per layer: ca_layernorm -> fused q|k|v ca_gemm (+bias) -> ca_attention (causal for text) ->
out_proj ca_gemm (+bias +residual) -> ca_layernorm -> fc1 ca_gemm (+bias + quick_gelu / gelu in the
epilogue) -> fc2 ca_gemm (+bias +residual).  Embedding lookups, the patch unfold and the class-token concat
are device-side torch indexing (plumbing).  No CPU fallback.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import torch
from torch import nn

from . import kernels as K
from .layers import HipLayerNorm, HipLinear, WeightArena, _f32, pack_concat_bias, pack_concat_rows

_ACT = {"quick_gelu": K.ACT_QUICK_GELU, "gelu": K.ACT_GELU}


class _SelfAttn(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.q_proj, self.k_proj, self.v_proj, self.out_proj = (HipLinear(c, c) for _ in range(4))
        self.wqkv = self.bqkv = None

    def pack(self, arena: WeightArena, dtype):
        mods = [self.q_proj, self.k_proj, self.v_proj]
        self.wqkv = pack_concat_rows(arena, dtype, mods)
        self.bqkv = pack_concat_bias(arena, mods)
        self.out_proj.pack(arena, dtype)


class _MLP(nn.Module):
    def __init__(self, c: int, inter: int):
        super().__init__()
        self.fc1, self.fc2 = HipLinear(c, inter), HipLinear(inter, c)


class CLIPEncoderLayer(nn.Module):
    def __init__(self, c: int, inter: int, heads: int, act: str, eps: float):
        super().__init__()
        self.heads, self.act = heads, _ACT[act]
        self.self_attn = _SelfAttn(c)
        self.layer_norm1 = HipLayerNorm(c, eps)
        self.mlp = _MLP(c, inter)
        self.layer_norm2 = HipLayerNorm(c, eps)

    def run(self, x: torch.Tensor, batch: int, tokens: int, causal: bool, key_mask=None) -> torch.Tensor:
        a = self.self_attn
        qkv = K.gemm(self.layer_norm1.run(x), a.wqkv.t, bias=a.bqkv.t)
        o = K.attention_spatial(qkv, batch, tokens, self.heads, causal=causal, key_mask=key_mask)
        x = a.out_proj.run(o, residual=x)
        h = self.mlp.fc1.run(self.layer_norm2.run(x), act=self.act)
        return self.mlp.fc2.run(h, residual=x)


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layers = nn.ModuleList([CLIPEncoderLayer(cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.hidden_act,
                                                      cfg.layer_norm_eps) for _ in range(cfg.num_hidden_layers)])


class _PackedModel(nn.Module):
    """prepare()/dtype/device plumbing shared by the two encoders (same contract as the UNet's)."""

    def _init_exec(self):
        self.arena: Optional[WeightArena] = None
        self.act_dtype = torch.float16

    @property
    def dtype(self):
        return self.act_dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def half(self):
        self.act_dtype, self.arena = torch.float16, None
        return self

    def bfloat16(self):
        self.act_dtype, self.arena = torch.bfloat16, None
        return self

    def to(self, *args, **kw):
        for a in list(args) + list(kw.values()):
            if a in (torch.float16, torch.bfloat16):
                self.act_dtype = a
        args = tuple(a for a in args if not isinstance(a, torch.dtype))
        kw = {k: v for k, v in kw.items() if not isinstance(v, torch.dtype)}
        if args or kw:
            super().to(*args, **kw)
            self.arena = None
        return self

    def prepare(self, device=None, dtype: Optional[torch.dtype] = None):
        if dtype is not None:
            self.act_dtype = dtype
        device = torch.device(device if device is not None else self.device)
        if device.type != "cuda":
            raise RuntimeError("prepare() needs a HIP device: the execution path has no CPU fallback")
        K.lib()
        arena = WeightArena()

        def walk(m):
            if hasattr(m, "pack"):
                m.pack(arena, self.act_dtype)
                return
            for ch in m.children():
                walk(ch)

        for ch in self.children():
            walk(ch)
        self._pack_extra(arena)
        arena.finalize(device)
        self.arena = arena
        return self

    def _pack_extra(self, arena):
        pass

    def _ready(self, device):
        if self.arena is None or self.arena.buffer.device != device:
            self.prepare(device)


class _Out(tuple):
    """transformers-style output: attribute access plus [0] indexing."""

    def __new__(cls, **kw):
        self = super().__new__(cls, tuple(v for v in kw.values() if v is not None))
        self.__dict__.update(kw)
        return self


# ------------------------------------------------------------------------------------------ text
class _TextEmbeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.token_embedding = nn.Embedding(cfg.vocab_size, cfg.hidden_size)
        self.position_embedding = nn.Embedding(cfg.max_position_embeddings, cfg.hidden_size)


class _TextTransformer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings = _TextEmbeddings(cfg)
        self.encoder = _Encoder(cfg)
        self.final_layer_norm = HipLayerNorm(cfg.hidden_size, cfg.layer_norm_eps)


TEXT_CONFIG = dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                   max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5, eos_token_id=2)


class CLIPTextModel(_PackedModel):
    def __init__(self, **config):
        super().__init__()
        cfg = dict(TEXT_CONFIG)
        cfg.update(config)
        self.config = SimpleNamespace(**cfg)
        self.text_model = _TextTransformer(self.config)
        self._init_exec()
        self._register_load_state_dict_pre_hook(self._prefix_keys)

    @classmethod
    def from_config(cls, config: Optional[dict] = None, **kw):
        cfg = dict(config or {})
        cfg.update(kw)
        return cls(**{k: v for k, v in cfg.items() if k in TEXT_CONFIG})

    @staticmethod
    def _prefix_keys(state_dict, prefix, *_):
        for k in list(state_dict.keys()):
            if k.endswith("position_ids"):
                state_dict.pop(k)
            elif not k.startswith("text_model."):  # transformers 5.x state_dict() has no wrapper level
                state_dict["text_model." + k] = state_dict.pop(k)

    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor, attention_mask=None, position_ids=None, output_attentions=None,
                output_hidden_states=None, return_dict=None):
        # attention_mask [B, L] (1 = attend): transformers adds it, expanded to [B,1,1,L], to the causal mask -- a masked token
        # is invisible as a KEY to every query.  Compel's default down-weighting (DownweightMode.MASK) passes one.
        key_mask = None
        if attention_mask is not None and not bool(torch.as_tensor(attention_mask).bool().all()):
            key_mask = torch.as_tensor(attention_mask).bool()
        if position_ids is not None:
            raise NotImplementedError("custom position_ids")
        dev = input_ids.device if input_ids.is_cuda else self.device
        self._ready(dev)
        ids = input_ids.to(dev)
        b, n = ids.shape
        if n > self.config.max_position_embeddings:
            raise ValueError(f"sequence length {n} exceeds {self.config.max_position_embeddings}")
        tm = self.text_model
        x = tm.embeddings.token_embedding.weight[ids] + tm.embeddings.position_embedding.weight[:n]
        x = x.to(self.act_dtype).reshape(b * n, -1).contiguous()
        hidden = [x.view(b, n, -1)] if output_hidden_states else None
        if key_mask is not None:
            if tuple(key_mask.shape) != (b, n):
                raise ValueError(f"attention_mask {tuple(key_mask.shape)} does not match input_ids {(b, n)}")
            key_mask = key_mask.to(device=dev, dtype=torch.uint8).contiguous()
        for layer in tm.encoder.layers:
            x = layer.run(x, b, n, causal=True, key_mask=key_mask)
            if hidden is not None:
                hidden.append(x.view(b, n, -1))
        last = tm.final_layer_norm.run(x).view(b, n, -1)
        if self.config.eos_token_id == 2:  # legacy configs (SD1.5): the EOS token has the highest id
            pos = ids.argmax(-1)
        else:
            pos = (ids == self.config.eos_token_id).int().argmax(-1)
        pooled = last[torch.arange(b, device=dev), pos]
        return _Out(last_hidden_state=last, pooler_output=pooled, hidden_states=None if hidden is None else tuple(hidden))


# ------------------------------------------------------------------------------------------ vision
class _PatchEmbedding(nn.Module):
    """Conv2d(3, C, P, stride P, bias=False) executed as a GEMM over unfolded patches (K padded to a multiple of 8)."""

    def __init__(self, c: int, patch: int):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(c, 3, patch, patch) * (3 * patch * patch) ** -0.5)
        self.k = 3 * patch * patch
        self.kp = (self.k + 7) // 8 * 8
        self.w = None

    def pack(self, arena: WeightArena, dtype):
        c = self.weight.shape[0]
        self.w = arena.add((c, self.kp), dtype, lambda: torch.nn.functional.pad(_f32(self.weight).reshape(c, self.k), (0, self.kp - self.k)))


class _VisionEmbeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.class_embedding = nn.Parameter(torch.randn(cfg.hidden_size) * 0.3)
        self.patch_embedding = _PatchEmbedding(cfg.hidden_size, cfg.patch_size)
        self.position_embedding = nn.Embedding((cfg.image_size // cfg.patch_size) ** 2 + 1, cfg.hidden_size)


class _VisionTransformer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings = _VisionEmbeddings(cfg)
        self.pre_layrnorm = HipLayerNorm(cfg.hidden_size, cfg.layer_norm_eps)  # (sic: the checkpoint key)
        self.encoder = _Encoder(cfg)
        self.post_layernorm = HipLayerNorm(cfg.hidden_size, cfg.layer_norm_eps)


VISION_CONFIG = dict(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=16, image_size=224,
                     patch_size=14, projection_dim=1024, hidden_act="gelu", layer_norm_eps=1e-5)  # laion CLIP-ViT-H-14 (IP-Adapter)


class CLIPVisionModelWithProjection(_PackedModel):
    def __init__(self, **config):
        super().__init__()
        cfg = dict(VISION_CONFIG)
        cfg.update(config)
        self.config = SimpleNamespace(**cfg)
        self.vision_model = _VisionTransformer(self.config)
        self.visual_projection = HipLinear(self.config.hidden_size, self.config.projection_dim, bias=False)
        self._init_exec()
        self._register_load_state_dict_pre_hook(lambda sd, *_: [sd.pop(k) for k in list(sd) if k.endswith("position_ids")])

    @classmethod
    def from_config(cls, config: Optional[dict] = None, **kw):
        cfg = dict(config or {})
        cfg.update(kw)
        return cls(**{k: v for k, v in cfg.items() if k in VISION_CONFIG})

    @torch.no_grad()
    def forward(self, pixel_values: torch.Tensor, output_attentions=None, output_hidden_states=None, return_dict=None):
        cfg = self.config
        if pixel_values.dim() != 4 or pixel_values.shape[1] != 3 or pixel_values.shape[2] != cfg.image_size or pixel_values.shape[3] != cfg.image_size:
            raise ValueError(f"expected [B,3,{cfg.image_size},{cfg.image_size}], got {tuple(pixel_values.shape)}")
        dev = pixel_values.device if pixel_values.is_cuda else self.device
        self._ready(dev)
        vm = self.vision_model
        b, p, g = pixel_values.shape[0], cfg.patch_size, cfg.image_size // cfg.patch_size
        pe = vm.embeddings.patch_embedding
        # [B,3,g,P,g,P] -> [B*g*g, 3*P*P] rows in (c, ph, pw) order = the conv weight's flattening
        patches = pixel_values.to(dev).reshape(b, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(b * g * g, pe.k)
        patches = torch.nn.functional.pad(patches, (0, pe.kp - pe.k)).to(self.act_dtype).contiguous()
        tok = K.gemm(patches, pe.w.t).view(b, g * g, -1)
        cls = vm.embeddings.class_embedding.to(self.act_dtype).expand(b, 1, -1)
        n = g * g + 1
        x = (torch.cat([cls, tok], 1).float() + vm.embeddings.position_embedding.weight).to(self.act_dtype).reshape(b * n, -1).contiguous()
        x = vm.pre_layrnorm.run(x)
        for layer in vm.encoder.layers:
            x = layer.run(x, b, n, causal=False)
        last = x.view(b, n, -1)
        pooled = vm.post_layernorm.run(last[:, 0].contiguous())
        embeds = self.visual_projection.run(pooled)
        return _Out(image_embeds=embeds, last_hidden_state=last)


# ------------------------------------------------------------------------------------------ host-side preprocessing
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def clip_preprocess(images, size: int = 224) -> torch.Tensor:
    """What transformers.CLIPImageProcessor() does with its defaults (modules/ip_adapter.py:75,192): resize the
    shorter side to `size` (bicubic), centre crop, scale to [0,1], normalise.  PIL images -> [B,3,size,size] fp32."""
    import numpy as np
    from PIL import Image
    if not isinstance(images, (list, tuple)):
        images = [images]
    out = []
    for im in images:
        im = im.convert("RGB")
        w, h = im.size
        short, long = (w, h) if w <= h else (h, w)
        ns, nl = size, int(size * long / short)
        im = im.resize((ns, nl) if w <= h else (nl, ns), resample=Image.BICUBIC)
        w, h = im.size
        left, top = (w - size) // 2, (h - size) // 2
        im = im.crop((left, top, left + size, top + size))
        a = np.asarray(im, dtype=np.float32) / 255.0
        a = (a - np.asarray(CLIP_MEAN, np.float32)) / np.asarray(CLIP_STD, np.float32)
        out.append(torch.from_numpy(a).permute(2, 0, 1))
    return torch.stack(out)
