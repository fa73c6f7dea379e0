"""Attention module + attention processors on the HIP kernels.

Keeps the reference's processor protocol (modules/attention_processor.py:395-402):
    processor(attn, hidden_states[B,N,C], encoder_hidden_states=None, attention_mask=None, temb=None)
with `attn` exposing to_q/to_k/to_v/to_out/heads/scale, and the same three processor kinds:
    AttnProcessor2_0      plain SDPA attention                        (reference :186-272)
    IPAttnProcessor2_0    + IP-Adapter K/V over the last 4 context tokens, out += scale*ip (:367-492)
    CNAttnProcessor2_0    ControlNet variant that drops the last 4 context tokens          (:561-646)
What differs is HOW they run: fused q|k|v (or k|v) GEMM -> flash attention kernel -> output
projection with the residual add fused into its epilogue; text K/V are computed once per b (the
reference repeats the prompt per frame, animatediff/models/attention.py:125) and cached across
denoising steps while the caller keeps passing the same prompt tensor.
`hidden_states` are device tensors in the activation dtype (bf16/fp16); there is no CPU path.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import kernels as K
from .context import dispatch
from .layers import HipLinear, LnFold, WeightArena, ln_fold_enabled, pack_concat_rows


class AttnProcessor2_0(nn.Module):
    """Default processor (self-, cross- and temporal attention)."""

    num_tokens = 0       # context tokens reserved for IP-Adapter
    drop_ip_tokens = False

    def __init__(self, hidden_size=None, cross_attention_dim=None):
        super().__init__()

    def ip_branch(self, attn, q, ctx, out, images, tokens, kv_rows, frames_per_kv, kv_mod, cache):
        return out

    def ip_fragments(self, attn, ctx, cache):
        """(packed K / V fragments of the image-prompt tokens, their count, scale) for the one-launch form, or None: no such tokens."""
        return None

    def __call__(self, attn: "Attention", hidden_states: torch.Tensor, encoder_hidden_states: Optional[torch.Tensor] = None,
                 attention_mask=None, temb=None, *, residual: Optional[torch.Tensor] = None, frames_per_kv: int = 1,
                 kv_mod: int = 0, temporal=None, cache: Optional[dict] = None, ln=None, row_sums: bool = False) -> torch.Tensor:
        """ln = (row statistics, LnFold): `hidden_states` is then the UN-normalised tensor and the LayerNorm that
        precedes this attention is folded into the q / q|k|v projection (built-in processors only)."""
        if attention_mask is not None:
            raise NotImplementedError("attention_mask is not used anywhere on the reference's path")
        B, N, C = hidden_states.shape
        x = hidden_states.reshape(B * N, C)
        res = None if residual is None else residual.reshape(B * N, C)
        if ln is not None and encoder_hidden_states is None and temporal is not None and getattr(attn, "tfrag", None) is not None:
            # LayerNorm + positional encoding + q|k|v + attention over the frames in one launch where the library takes the shape
            b, f, n = temporal
            ofrag = getattr(attn, "ofrag", None)
            if ofrag is not None and dispatch.attn_out_fused:  # ... and the output projection + bias + residual in the same launch (ABI v12)
                to_out = attn.to_out[0]
                out = K.tattn_fused(x, attn.tfrag.t, attn.tgamma.t, attn.tbias_pe.t, b, f, n, attn.heads, attn.tln_eps, attn.scale,
                                    w_out_frag=ofrag.t, bias_out=None if to_out.b is None else to_out.b.t, residual=res)
                if out is not None:
                    return out.reshape(B, N, C)
            o = K.tattn_fused(x, attn.tfrag.t, attn.tgamma.t, attn.tbias_pe.t, b, f, n, attn.heads, attn.tln_eps, attn.scale)
            if o is not None:
                out = attn.to_out[0].run(o, residual=res, row_sums=row_sums)
                return K.carry_row_sums(out.reshape(B, N, C), out)
        builtin = type(self).ip_branch is AttnProcessor2_0.ip_branch
        with_ip = (not builtin and type(self).ip_fragments is not AttnProcessor2_0.ip_fragments and dispatch.attn_out_fused and dispatch.xattn_ip_fused
                   and getattr(attn, "ofrag", None) is not None)
        if (ln is not None and encoder_hidden_states is not None and cache is not None and getattr(attn, "xfrag", None) is not None
                and (builtin or with_ip)):
            # LayerNorm + to_q + attention over the text tokens in one launch where the library takes the shape.  The K / V fragments are
            # packed once per window beside the projected K / V (refresh_window_caches repacks them in place).  A processor with a second
            # attention over image-prompt tokens (the IP-Adapter's) rides along only in the form that ends with the output projection
            # (ABI v13: its q never leaves the kernel); otherwise it keeps the separate launches.
            ctx = encoder_hidden_states
            nb, L, cd = ctx.shape
            nk = L - self.num_tokens
            kv = cache.get(("kv", id(attn)))
            if kv is None:
                kv = cache[("kv", id(attn))] = K.gemm(ctx.reshape(nb * L, cd), attn.kv.t)
            ent = cache.get(("kvf", id(attn)))
            if ent is None:
                ent = cache[("kvf", id(attn))] = (K.xattn_pack_kv(kv, nb, L, nk, attn.scale), nk, L)
            if ent[0] is not None and ent[1] == nk:
                ofrag = getattr(attn, "ofrag", None)
                if ofrag is not None and dispatch.attn_out_fused:  # ... and to_out + bias + residual in the same launch (ABI v12)
                    to_out = attn.to_out[0]
                    ipf = self.ip_fragments(attn, ctx, cache) if with_ip else None
                    if not with_ip or ipf is not None:
                        out = K.xattn_fused(x, attn.xfrag.t, ln[1].b.t, ent[0], B, N, frames_per_kv, kv_mod, nk, ln[1].eps, w_out_frag=ofrag.t,
                                            bias_out=None if to_out.b is None else to_out.b.t, residual=res,
                                            kv_frag_ip=None if ipf is None else ipf[0], nk_ip=0 if ipf is None else ipf[1],
                                            ip_scale=1.0 if ipf is None else ipf[2])
                        if out is not None:
                            return out.reshape(B, N, C)
                o = K.xattn_fused(x, attn.xfrag.t, ln[1].b.t, ent[0], B, N, frames_per_kv, kv_mod, nk, ln[1].eps) if builtin else None
                if o is not None:
                    out = attn.to_out[0].run(o, residual=res, row_sums=row_sums)
                    return K.carry_row_sums(out.reshape(B, N, C), out)
        if ln is not None:
            st, fold = ln
            if encoder_hidden_states is None and temporal is not None and fold.pe is not None:
                b, f, n = temporal
                qkv = K.gemm(x, fold.w.t, bias=fold.b.t, ln=(st, fold.cs.t), rowbias=fold.rowbias(b, f), rows_per_group=n)
            else:
                qkv = K.gemm(x, fold.w.t, bias=fold.b.t, ln=(st, fold.cs.t))
        if encoder_hidden_states is None:
            if ln is None:
                qkv = K.gemm(x, attn.qkv.t)
            if temporal is not None:
                b, f, n = temporal
                o = K.attention_temporal(qkv, b, f, n, attn.heads)
            else:
                o = K.attention_spatial(qkv, B, N, attn.heads)
        else:
            ctx = encoder_hidden_states
            nb, L, cd = ctx.shape
            q = qkv if ln is not None else K.gemm(x, attn.to_q.w.t)
            kv = None if cache is None else cache.get(("kv", id(attn)))
            if kv is None:
                kv = K.gemm(ctx.reshape(nb * L, cd), attn.kv.t)
                if cache is not None:
                    cache[("kv", id(attn))] = kv
            nk = L - self.num_tokens
            o = K.attention_cross(q, kv, B, N, attn.heads, nk, L, frames_per_kv, kv_mod=kv_mod)
            o = self.ip_branch(attn, q, ctx, o, B, N, L, frames_per_kv, kv_mod, cache)
        # row_sums: the output feeds a folded LayerNorm next -- its statistics come out of this GEMM's epilogue where it can
        out = attn.to_out[0].run(o, residual=res, row_sums=row_sums)
        return K.carry_row_sums(out.reshape(B, N, C), out)


class IPAttnProcessor2_0(AttnProcessor2_0):
    """IP-Adapter processor: extra K/V projections for the image-prompt tokens."""

    def __init__(self, hidden_size, cross_attention_dim=None, scale=1.0, num_tokens=4):
        super().__init__()
        self.hidden_size = hidden_size
        self.cross_attention_dim = cross_attention_dim
        self.scale = scale
        self.num_tokens = num_tokens
        self.to_k_ip = HipLinear(cross_attention_dim or hidden_size, hidden_size, bias=False)
        self.to_v_ip = HipLinear(cross_attention_dim or hidden_size, hidden_size, bias=False)
        self.kv_ip = None

    def pack(self, arena: WeightArena, dtype):
        self.kv_ip = pack_concat_rows(arena, dtype, [self.to_k_ip, self.to_v_ip])

    def _kv_ip(self, ctx, cache):
        if self.kv_ip is None:
            raise RuntimeError("IPAttnProcessor2_0 installed after prepare(); call model.prepare() again")
        nb, L, cd = ctx.shape
        kvip = None if cache is None else cache.get(("kv_ip", id(self)))
        if kvip is None:
            kvip = K.gemm(ctx.reshape(nb * L, cd), self.kv_ip.t)  # all rows; only the last num_tokens are read
            if cache is not None:
                cache[("kv_ip", id(self))] = kvip
        return kvip

    def ip_fragments(self, attn, ctx, cache):
        """The to_k_ip | to_v_ip projection of the image-prompt tokens as the MFMA fragments K.xattn_fused reads (packed once per window;
        refresh_window_caches repacks in place), or None for shapes the one-launch form does not take."""
        nb, L, cd = ctx.shape
        if cache is None or not (1 <= self.num_tokens <= 16):
            return None
        ent = cache.get(("kvf_ip", id(self)))
        if ent is None:
            kvip = self._kv_ip(ctx, cache)
            ent = cache[("kvf_ip", id(self))] = (K.xattn_pack_kv(kvip, nb, L, self.num_tokens, attn.scale, row_offset=L - self.num_tokens),
                                                 self.num_tokens, L, attn.scale)
        if ent[0] is None or ent[1] != self.num_tokens or ent[2] != L:
            return None
        return ent[0], self.num_tokens, float(self.scale)

    def ip_branch(self, attn, q, ctx, out, images, tokens, kv_rows, frames_per_kv, kv_mod, cache):
        nb, L, cd = ctx.shape
        kvip = self._kv_ip(ctx, cache)
        return K.attention_cross(q, kvip, images, tokens, attn.heads, self.num_tokens, L, frames_per_kv, out=out,
                                 out_scale=float(self.scale), accumulate=True, kv_row_offset=L - self.num_tokens,
                                 kv_mod=kv_mod)


class CNAttnProcessor2_0(AttnProcessor2_0):
    """ControlNet processor under IP-Adapter: only the text part of the context is used."""

    def __init__(self, num_tokens=4):
        super().__init__()
        self.num_tokens = num_tokens


# reference aliases (modules/ip_adapter.py:22-25 picks the 2_0 classes when torch has SDPA)
AttnProcessor = AttnProcessor2_0
IPAttnProcessor = IPAttnProcessor2_0
CNAttnProcessor = CNAttnProcessor2_0


class Attention(nn.Module):
    """diffusers-style Attention block: bias-free q/k/v projections, biased output projection."""

    def __init__(self, query_dim: int, cross_attention_dim: Optional[int] = None, heads: int = 8, dim_head: int = 64,
                 dropout: float = 0.0, bias: bool = False, upcast_attention: bool = False, processor=None, **_):
        super().__init__()
        inner = heads * dim_head
        self.is_cross = cross_attention_dim is not None
        ctx_dim = cross_attention_dim if self.is_cross else query_dim
        self.query_dim, self.cross_attention_dim, self.inner_dim = query_dim, ctx_dim, inner
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0
        self.to_q = HipLinear(query_dim, inner, bias=bias)
        self.to_k = HipLinear(ctx_dim, inner, bias=bias)
        self.to_v = HipLinear(ctx_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([HipLinear(inner, query_dim, bias=True), nn.Identity()])
        self.processor = processor if processor is not None else AttnProcessor2_0()
        self.qkv = self.kv = self.fold = None

    def set_processor(self, processor, _remove_lora: bool = False):
        if isinstance(getattr(self, "processor", None), nn.Module) and not isinstance(processor, nn.Module):
            self._modules.pop("processor", None)
        self.processor = processor

    def get_processor(self, return_deprecated_lora: bool = False):
        return self.processor

    def builtin_processor(self) -> bool:
        return type(self.processor) in (AttnProcessor2_0, IPAttnProcessor2_0, CNAttnProcessor2_0)

    def pack(self, arena: WeightArena, dtype, fold_ln=None, pe=None):
        """fold_ln: the LayerNorm in front of this attention; with a built-in processor it is folded into the
        query-side projection (LnFold) and the plain copies of those weights are not packed."""
        self.fold = None
        if fold_ln is not None and self.builtin_processor() and ln_fold_enabled():
            self.fold = LnFold(arena, dtype, fold_ln, [self.to_q] if self.is_cross else [self.to_q, self.to_k, self.to_v], pe=pe)
            if self.is_cross:
                self.kv = pack_concat_rows(arena, dtype, [self.to_k, self.to_v])
                # the one-launch form of LayerNorm + to_q + attention over the text tokens (K.xattn_fused: C = 320, 8 heads of 40)
                self.xfrag = None
                if (self.heads, self.inner_dim, self.query_dim) == (8, 320, 320):
                    from .layers import frag_order_xattn
                    fold = self.fold
                    self.xfrag = arena.add((122880,), dtype, lambda: frag_order_xattn(fold.w_fold().to(dtype).float()))
        elif self.is_cross:
            self.to_q.pack(arena, dtype)
            self.kv = pack_concat_rows(arena, dtype, [self.to_k, self.to_v])
        else:
            self.qkv = pack_concat_rows(arena, dtype, [self.to_q, self.to_k, self.to_v])
        self.to_out[0].pack(arena, dtype)
        # the output projection as the last stage of the one-launch attentions (ABI v12: K.tattn_fused / K.xattn_fused with w_out_frag)
        self.ofrag = None
        if (self.heads, self.inner_dim, self.query_dim) == (8, 320, 320) and self.fold is not None:
            from .layers import _f32, frag_order_wout
            to_out = self.to_out[0]
            self.ofrag = arena.add((102400,), dtype, lambda: frag_order_wout(_f32(to_out.weight)))
        if hasattr(self.processor, "pack"):
            self.processor.pack(arena, dtype)

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kw)
