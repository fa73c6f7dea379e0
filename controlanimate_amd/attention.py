"""Spatial transformer (Transformer3DModel / BasicTransformerBlock / FeedForward) on HIP kernels.

Mirrors the reference's animatediff/models/attention.py (Transformer3DModel :52-167,
BasicTransformerBlock :170-300, FeedForward :303-357) in names and checkpoint keys.  Execution:
  GroupNorm(eps 1e-6) -> proj_in (1x1 conv = row GEMM in NHWC) ->
  [LayerNorm -> fused q|k|v GEMM -> flash self-attention -> out GEMM (+residual)]
  [LayerNorm -> q GEMM, cached text k|v -> cross attention (+IP-Adapter branch) -> out GEMM (+residual)]
  [LayerNorm -> GEGLU GEMM (value*gelu(gate) fused in the epilogue) -> out GEMM (+residual)]
  -> proj_out GEMM (+ block input residual).
The rearranges of the reference ('b c f h w -> (b f) c h w', permute to tokens) do not exist here:
NHWC activations already are [tokens, channels] rows.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import kernels as K
from .attention_processor import Attention, AttnProcessor2_0
from .context import ExecCtx, dispatch
from .layers import HipConv1x1, HipGroupNorm, HipLayerNorm, HipLinear, LnFold, WeightArena, _f32, geglu_interleave, ln_fold_enabled


class GEGLU(nn.Module):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.dim_in, self.dim_out = dim_in, dim_out
        self.proj = HipLinear(dim_in, dim_out * 2)
        self.w = self.b = None

    def pack(self, arena: WeightArena, dtype):
        self.w = arena.add((2 * self.dim_out, self.dim_in), dtype, lambda: geglu_interleave(_f32(self.proj.weight)))
        self.b = arena.add((2 * self.dim_out,), torch.float32, lambda: geglu_interleave(_f32(self.proj.bias)))

    def run(self, x: torch.Tensor) -> torch.Tensor:
        return K.gemm(x, self.w.t, bias=self.b.t, geglu=True)


class FeedForward(nn.Module):
    def __init__(self, dim: int, dim_out: Optional[int] = None, mult: int = 4, dropout: float = 0.0,
                 activation_fn: str = "geglu", final_dropout: bool = False):
        super().__init__()
        if activation_fn != "geglu":
            raise NotImplementedError(activation_fn)
        inner = int(dim * mult)
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Identity(), HipLinear(inner, dim_out or dim)])

    def pack(self, arena, dtype, fold_ln=None):
        """fold_ln: the LayerNorm in front of the feed-forward; it is folded into the GEGLU projection."""
        self.fold = None
        if fold_ln is not None and ln_fold_enabled():
            self.fold = LnFold(arena, dtype, fold_ln, [self.net[0].proj], geglu=True)
        else:
            self.net[0].pack(arena, dtype)
        self.net[2].pack(arena, dtype)
        # C = 320 (the 64x64-latent level): the output projection once more in the fragment order of the fused feed-forward
        # kernel (ca_ff_fused: GEGLU projection + output projection in one launch, the [M, 1280] intermediate stays in LDS)
        self.w2f = None
        out = self.net[2]
        if self.fold is not None and self.fold.w.frag is not None and (out.out_features, out.in_features) == (320, 1280):
            from .layers import frag_order2
            self.w2f = arena.add((320, 1280), dtype, lambda: frag_order2(out.weight.detach().float()))

    def run_with_proj_out(self, x: torch.Tensor, residual: Optional[torch.Tensor], proj_frag, proj_bias, proj_residual) -> Optional[torch.Tensor]:
        """The feed-forward AND the transformer's proj_out + bias + residual behind it in one launch (ca_ff_fused with w_out_frag,
        ABI v12) -- or None where the one-launch feed-forward does not apply (the caller runs the feed-forward, then proj_out)."""
        fold = getattr(self, "fold", None)
        if fold is None or getattr(self, "w2f", None) is None or proj_frag is None or not dispatch.attn_out_fused:
            return None
        return K.ff_fused(x, fold.w.frag[0].t, fold.b.t, fold.cs.t, self.w2f.t, None if self.net[2].b is None else self.net[2].b.t,
                          fold.eps, residual=residual, w_out_frag=proj_frag.t, bias_out=None if proj_bias is None else proj_bias.t,
                          residual_out=proj_residual)

    def run(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None, sums=None) -> torch.Tensor:
        """With a folded LayerNorm `x` is the UN-normalised input (`sums`: row sums its producer left, K.row_sums_of)."""
        fold = getattr(self, "fold", None)
        if fold is not None and getattr(self, "w2f", None) is not None and sums is None:
            y = K.ff_fused(x, fold.w.frag[0].t, fold.b.t, fold.cs.t, self.w2f.t, None if self.net[2].b is None else self.net[2].b.t,
                           fold.eps, residual=residual)
            if y is not None:
                return y
        if fold is not None:
            h = K.gemm(x, fold.w.t, bias=fold.b.t, geglu=True, ln=(K.RowStats(x, fold.eps, sums), fold.cs.t))
        else:
            h = self.net[0].run(x)
        return self.net[2].run(h, residual=residual)


class BasicTransformerBlock(nn.Module):
    def __init__(self, query_dim: int, num_attention_heads: int, attention_head_dim: int, dropout=0.0,
                 cross_attention_dim: Optional[int] = None, activation_fn: str = "geglu", attention_bias: bool = False,
                 upcast_attention: bool = False, unet_use_cross_frame_attention=False, unet_use_temporal_attention=False, **_):
        super().__init__()
        if unet_use_cross_frame_attention or unet_use_temporal_attention:
            raise NotImplementedError("both flags are false in configs/inference/inference-v{1,2}.yaml")
        # the reference's block subclasses diffusers Attention, which gives it a (dead) processor
        # slot of its own; kept so that `attn_processors` enumerates the same 88 / 90 keys.
        self.processor = AttnProcessor2_0()
        self.attn1 = Attention(query_dim, heads=num_attention_heads, dim_head=attention_head_dim, bias=attention_bias)
        self.norm1 = HipLayerNorm(query_dim)
        self.attn2 = Attention(query_dim, cross_attention_dim=cross_attention_dim, heads=num_attention_heads,
                               dim_head=attention_head_dim, bias=attention_bias) if cross_attention_dim is not None else None
        self.norm2 = HipLayerNorm(query_dim) if cross_attention_dim is not None else None
        self.ff = FeedForward(query_dim, activation_fn=activation_fn)
        self.norm3 = HipLayerNorm(query_dim)

    def get_processor(self, return_deprecated_lora: bool = False):
        return self.processor

    def set_processor(self, processor, _remove_lora: bool = False):
        if isinstance(getattr(self, "processor", None), nn.Module) and not isinstance(processor, nn.Module):
            self._modules.pop("processor", None)
        self.processor = processor

    def pack(self, arena, dtype):
        # The three LayerNorms are folded into the projections they feed (LnFold: gamma into the weights, mean /
        # rstd in the GEMM epilogue): the normalised copy of the residual stream is never written or re-read.
        # An attention with a user-supplied processor keeps the plain LayerNorm -> processor protocol.
        self.attn1.pack(arena, dtype, fold_ln=self.norm1)
        self.norm1.pack(arena, dtype)
        if self.attn2 is not None:
            self.attn2.pack(arena, dtype, fold_ln=self.norm2)
            self.norm2.pack(arena, dtype)
        self.ff.pack(arena, dtype, fold_ln=self.norm3)
        self.norm3.pack(arena, dtype)

    def forward(self, x: torch.Tensor, ctx: ExecCtx, proj_out=None) -> torch.Tensor:
        """x: [images, tokens, C].  proj_out = (fragment-ordered weight, bias, residual rows) of the transformer this block ends:
        applied in the feed-forward's launch where possible -- the result then carries `_proj_out_done`."""
        return self.forward_rest(self.forward_self(x), ctx, proj_out=proj_out)

    def forward_self(self, x: torch.Tensor) -> torch.Tensor:
        """norm1 + self-attention + residual: the part that does not see the prompt (UNet3DConditionModel.forward_nhwc
        runs it once for the two identical halves of a classifier-free-guidance batch)."""
        B, N, C = x.shape
        # (each projection that feeds a folded LayerNorm is asked for the row sums of its output: K.gemm(row_sums=True))
        next_folded = (self.attn2.fold if self.attn2 is not None else self.ff.fold) is not None
        if self.attn1.fold is not None:
            x = self.attn1(x, residual=x, ln=(K.RowStats(x.view(B * N, C), self.norm1.eps, K.row_sums_of(x)), self.attn1.fold),
                           row_sums=next_folded)
        else:
            x = self.attn1(self.norm1.run(x.view(B * N, C)).view(B, N, C), residual=x, row_sums=next_folded)
        return x

    def forward_rest(self, x: torch.Tensor, ctx: ExecCtx, proj_out=None) -> torch.Tensor:
        """norm2 + cross-attention, norm3 + feed-forward (with their residuals)."""
        B, N, C = x.shape
        if self.attn2 is not None:
            kw = dict(encoder_hidden_states=ctx.ehs, residual=x, frames_per_kv=ctx.frames_per_kv, kv_mod=ctx.kv_mod, cache=ctx.cache,
                      row_sums=self.ff.fold is not None)
            if self.attn2.fold is not None:
                x = self.attn2(x, ln=(K.RowStats(x.view(B * N, C), self.norm2.eps, K.row_sums_of(x)), self.attn2.fold), **kw)
            else:
                x = self.attn2(self.norm2.run(x.view(B * N, C)).view(B, N, C), **kw)
        x2 = x.view(B * N, C)
        if self.ff.fold is None:
            return self.ff.run(self.norm3.run(x2), residual=x2).view(B, N, C)
        if proj_out is not None and K.row_sums_of(x) is None:
            y = self.ff.run_with_proj_out(x2, x2, *proj_out)
            if y is not None:
                y = y.view(B, N, C)
                y._proj_out_done = True
                return y
        return self.ff.run(x2, residual=x2, sums=K.row_sums_of(x)).view(B, N, C)


class Transformer3DModel(nn.Module):
    def __init__(self, num_attention_heads: int = 16, attention_head_dim: int = 88, in_channels: Optional[int] = None,
                 num_layers: int = 1, norm_num_groups: int = 32, cross_attention_dim: Optional[int] = None,
                 use_linear_projection: bool = False, **kw):
        super().__init__()
        if use_linear_projection:
            raise NotImplementedError("SD1.5 uses 1x1-conv projections")
        inner = num_attention_heads * attention_head_dim
        self.in_channels = in_channels
        self.norm = HipGroupNorm(norm_num_groups, in_channels, eps=1e-6)
        self.proj_in = HipConv1x1(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, num_attention_heads, attention_head_dim, cross_attention_dim=cross_attention_dim, **kw)
            for _ in range(num_layers)])
        self.proj_out = HipConv1x1(inner, in_channels)

    def pack(self, arena, dtype):
        self.norm.pack(arena, dtype)
        self.proj_in.pack(arena, dtype)
        for b in self.transformer_blocks:
            b.pack(arena, dtype)
        self.proj_out.pack(arena, dtype)
        # proj_out once more in the fragment order of the output stage behind the last block's feed-forward (ca_ff_fused, ABI v12)
        self.proj_out_frag = None
        po = self.proj_out
        if tuple(po.weight.shape[:2]) == (320, 320) and getattr(self.transformer_blocks[-1].ff, "w2f", None) is not None:
            from .layers import frag_order_wout
            self.proj_out_frag = arena.add((102400,), dtype, lambda: frag_order_wout(_f32(po.weight).reshape(320, 320)))

    def forward(self, x: torch.Tensor, ctx: ExecCtx, shared_half: bool = False) -> torch.Tensor:
        """shared_half: `x` holds ONE of the two identical halves of a classifier-free-guidance batch (ctx describes the
        full batch): GroupNorm, proj_in and the first block's self-attention run once, then the activations are repeated
        and the prompt-dependent rest runs on the full batch.  Identical inputs give identical outputs, so the result is
        what the full-batch path computes."""
        images, h, w, c = x.shape
        rows = images * h * w
        y = self.norm.run(x)  # always per image (reference rearranges to (b f) first, attention.py:124)
        y0 = self.proj_in.run(y.view(rows, c), row_sums=self.transformer_blocks[0].attn1.fold is not None)
        y = K.carry_row_sums(y0.view(images, h * w, -1), y0)
        nblk = len(self.transformer_blocks)
        pfrag = getattr(self, "proj_out_frag", None)

        def proj(k, x_rows):  # the last block takes proj_out + bias + residual into its feed-forward's launch where it can
            return (pfrag, self.proj_out.b, x_rows) if (k == nblk - 1 and pfrag is not None) else None
        if shared_half:
            y = self.transformer_blocks[0].forward_self(y)
            y, x = K.repeat_batch(y), K.repeat_batch(x)  # (= torch.cat([t, t]), one read each)
            images, rows = 2 * images, 2 * rows
            y = self.transformer_blocks[0].forward_rest(y, ctx, proj_out=proj(0, x.view(rows, c)))
        for k in range(1 if shared_half else 0, nblk):
            y = self.transformer_blocks[k](y, ctx, proj_out=proj(k, x.view(rows, c)))
        if getattr(y, "_proj_out_done", False):
            return y.view(images, h, w, c)
        return self.proj_out.run(y.view(rows, -1), residual=x.view(rows, c)).view(images, h, w, c)
