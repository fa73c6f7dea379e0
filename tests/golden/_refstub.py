"""Container-only import shim for the reference (NOT shipped with the product, not used on the GPU box).

The reference (/root/reference) needs diffusers==0.23.0, omegaconf, compel, cv2, controlnet_aux,
torchvision, ... none of which are installed here and there is no network.  This module registers
stand-in modules so that the reference's OWN files (animatediff/models/*.py,
modules/attention_processor.py, the pipeline file's LCMScheduler / get_w_embedding) import and run
on CPU, which is how tests/golden/*.npz were generated (make_golden.py).

The only ARITHMETIC supplied here is a restatement of the handful of diffusers 0.23.0 classes the
reference's modules subclass or call (Attention projections, GEGLU, FeedForward, Timesteps,
TimestepEmbedding) -- see SURVEY.md App. A; everything else is the reference's own code.
Refuses to run when /root/reference is absent.
"""
from __future__ import annotations

import functools
import importlib.abc
import importlib.machinery
import inspect
import math
import os
import sys
import types

import torch
import torch.nn.functional as F
from torch import nn

REFERENCE = "/root/reference"


# ----------------------------------------------------------------------------- config plumbing
class FrozenDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def register_to_config(init):
    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        params = list(inspect.signature(init).parameters.items())[1:]
        cfg = {n: p.default for n, p in params if p.default is not inspect.Parameter.empty}
        for (n, _), a in zip(params, args):
            cfg[n] = a
        cfg.update(kwargs)
        self._internal_dict = FrozenDict(cfg)
        init(self, *args, **kwargs)
    return inner


class ConfigMixin:
    config_name = "config.json"

    @property
    def config(self):
        return self._internal_dict

    def register_to_config(self, **kw):
        d = dict(getattr(self, "_internal_dict", {}))
        d.update(kw)
        self._internal_dict = FrozenDict(d)

    @classmethod
    def from_config(cls, config, **kwargs):
        names = set(inspect.signature(cls.__init__).parameters) - {"self"}
        merged = {k: v for k, v in dict(config).items() if k in names}
        merged.update({k: v for k, v in kwargs.items() if k in names})
        return cls(**merged)


class ModelMixin(nn.Module):
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device


class BaseOutput:
    pass


class SchedulerMixin:
    pass


class _Logger:
    def __getattr__(self, name):
        return lambda *a, **k: None


class _Logging:
    @staticmethod
    def get_logger(name=None):
        return _Logger()


def deprecate(*a, **k):
    return None


# ----------------------------------------------------------------------------- diffusers arithmetic
class LoRACompatibleConv(nn.Conv2d):
    def forward(self, x, scale: float = 1.0):
        return super().forward(x)


class LoRACompatibleLinear(nn.Linear):
    def forward(self, x, scale: float = 1.0):
        return super().forward(x)


class LoRALinearLayer(nn.Module):
    pass


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels, self.flip, self.shift = num_channels, flip_sin_to_cos, downscale_freq_shift

    def forward(self, timesteps):
        half = self.num_channels // 2
        exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32, device=timesteps.device)
        exponent = exponent / (half - self.shift)
        emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
        emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
        if self.flip:
            emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
        return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None, post_act_fn=None, cond_proj_dim=None):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.cond_proj = nn.Linear(cond_proj_dim, in_channels, bias=False) if cond_proj_dim is not None else None
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, out_dim or time_embed_dim)

    def forward(self, sample, condition=None):
        if condition is not None:
            sample = sample + self.cond_proj(condition)
        return self.linear_2(self.act(self.linear_1(sample)))


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = LoRACompatibleLinear(dim_in, dim_out * 2)

    def forward(self, hidden_states, scale: float = 1.0):
        hidden_states, gate = self.proj(hidden_states, scale).chunk(2, dim=-1)
        return hidden_states * F.gelu(gate)


class GELU(nn.Module):
    pass


class ApproximateGELU(nn.Module):
    pass


class AdaLayerNorm(nn.Module):
    pass


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", final_dropout=False):
        super().__init__()
        assert activation_fn == "geglu"
        inner = int(dim * mult)
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(dropout), LoRACompatibleLinear(inner, dim_out or dim)])

    def forward(self, hidden_states, scale: float = 1.0):
        for m in self.net:
            hidden_states = m(hidden_states)
        return hidden_states


class Attention(nn.Module):
    """diffusers 0.23.0 Attention: projections + processor protocol."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, upcast_softmax=False, cross_attention_norm=None, added_kv_proj_dim=None,
                 norm_num_groups=None, out_bias=True, scale_qk=True, only_cross_attention=False, eps=1e-5,
                 rescale_output_factor=1.0, residual_connection=False, processor=None, **_):
        super().__init__()
        inner = dim_head * heads
        ctx = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.cross_attention_dim = ctx
        self.upcast_attention, self.upcast_softmax = upcast_attention, upcast_softmax
        self.rescale_output_factor, self.residual_connection = rescale_output_factor, residual_connection
        self.scale = dim_head ** -0.5 if scale_qk else 1.0
        self.heads = heads
        self.sliceable_head_dim = heads
        self.added_kv_proj_dim = added_kv_proj_dim
        self.only_cross_attention = only_cross_attention
        self.group_norm = None
        self.spatial_norm = None
        self.norm_cross = None
        self.to_q = LoRACompatibleLinear(query_dim, inner, bias=bias)
        self.to_k = LoRACompatibleLinear(ctx, inner, bias=bias)
        self.to_v = LoRACompatibleLinear(ctx, inner, bias=bias)
        self.to_out = nn.ModuleList([LoRACompatibleLinear(inner, query_dim, bias=out_bias), nn.Dropout(dropout)])
        if processor is None:
            from modules.attention_processor import AttnProcessor2_0  # the REFERENCE's own processor
            processor = AttnProcessor2_0()
        self.processor = processor

    def set_processor(self, processor, _remove_lora=False):
        if hasattr(self, "processor") and isinstance(self.processor, nn.Module) and not isinstance(processor, nn.Module):
            self._modules.pop("processor")
        self.processor = processor

    def get_processor(self, return_deprecated_lora=False):
        return self.processor

    def prepare_attention_mask(self, attention_mask, target_length, batch_size, out_dim=3):
        return attention_mask

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kw)


class AttnProcessor:
    pass


class AttnAddedKVProcessor:
    pass


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    gdev = generator.device.type if generator is not None else "cpu"
    t = torch.randn(shape, generator=generator, device=gdev, dtype=dtype)
    return t.to(device) if device is not None else t


# ----------------------------------------------------------------------------- fake module tree
class _Auto(types.ModuleType):
    """A module whose unknown attributes are fresh dummy classes (name-only stand-ins)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (), {"__init__": lambda self, *a, **k: None})
        setattr(self, name, cls)
        return cls


_AUTO_ROOTS = ("diffusers", "torchvision", "cv2", "controlnet_aux", "imageio", "color_matcher", "compel",
               "omegaconf", "xformers", "realesrgan", "basicsr", "gfpgan")


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _AUTO_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Auto(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        _populate(module)


def _populate(m):
    n = m.__name__
    table = {
        "diffusers": dict(SchedulerMixin=SchedulerMixin, ConfigMixin=ConfigMixin),
        "diffusers.configuration_utils": dict(ConfigMixin=ConfigMixin, register_to_config=register_to_config, FrozenDict=FrozenDict),
        "diffusers.models.modeling_utils": dict(ModelMixin=ModelMixin),
        "diffusers.utils": dict(BaseOutput=BaseOutput, USE_PEFT_BACKEND=False, deprecate=deprecate, logging=_Logging,
                                WEIGHTS_NAME="diffusion_pytorch_model.bin",
                                SAFETENSORS_WEIGHTS_NAME="diffusion_pytorch_model.safetensors",
                                is_accelerate_available=lambda: False),
        "diffusers.utils.import_utils": dict(is_xformers_available=lambda: False),
        "diffusers.utils.torch_utils": dict(maybe_allow_in_graph=lambda c: c, randn_tensor=randn_tensor),
        "diffusers.models.embeddings": dict(TimestepEmbedding=TimestepEmbedding, Timesteps=Timesteps),
        "diffusers.models.attention_processor": dict(Attention=Attention, AttnProcessor=AttnProcessor,
                                                     AttnAddedKVProcessor=AttnAddedKVProcessor,
                                                     AttentionProcessor=object, ADDED_KV_ATTENTION_PROCESSORS=(),
                                                     CROSS_ATTENTION_PROCESSORS=(), __all__=[]),
        "diffusers.models.attention": dict(Attention=Attention, AdaLayerNorm=AdaLayerNorm, FeedForward=FeedForward),
        "diffusers.models.activations": dict(GEGLU=GEGLU, GELU=GELU, ApproximateGELU=ApproximateGELU),
        "diffusers.models.lora": dict(LoRACompatibleConv=LoRACompatibleConv, LoRACompatibleLinear=LoRACompatibleLinear,
                                      LoRALinearLayer=LoRALinearLayer, adjust_lora_scale_text_encoder=lambda *a, **k: None),
        "diffusers.schedulers.scheduling_utils": dict(SchedulerMixin=SchedulerMixin),
    }
    for k, v in table.get(n, {}).items():
        setattr(m, k, v)


_installed = False


def install():
    """Makes `import animatediff...` / `import modules...` (the reference) work in this container."""
    global _installed
    if _installed:
        return
    if not os.path.isdir(REFERENCE):
        raise RuntimeError(f"{REFERENCE} is absent: golden generation only runs in the build container")
    import transformers  # noqa: F401  (must be imported before a fake torchvision exists)
    try:
        from transformers import CLIPTextModel, CLIPTokenizer  # noqa: F401
    except Exception:
        pass
    sys.dont_write_bytecode = True  # never litter /root/reference with __pycache__
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, REFERENCE)
    _installed = True
