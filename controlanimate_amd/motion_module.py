"""AnimateDiff motion module (temporal transformer) on HIP kernels.

Mirrors the reference's animatediff/models/motion_module.py in names and checkpoint keys
(VanillaTemporalModule :50-84, TemporalTransformer3DModel :87-160, TemporalTransformerBlock
:163-224, PositionalEncoding :227-245, VersatileAttention :248-344).  Differences in execution:
  * no '(b f) d c -> (b d) f c' transposes: the attention kernel walks the frame axis with a
    row stride of tokens*C inside the (b f n) row order (SURVEY 2.1 "Temporal self-attention");
  * `x + pe[:, :f]` is fused into the preceding LayerNorm kernel (rows know their frame);
  * q|k|v is ONE GEMM and is computed once (the reference computes q, k, v, discards them and lets
    the processor recompute, motion_module.py:299-311 -- no effect on outputs, not reproduced);
  * residual adds live in the GEMM epilogues.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from . import kernels as K
from .attention import FeedForward
from .attention_processor import Attention
from .context import ExecCtx
from .layers import HipGroupNorm, HipLayerNorm, HipLinear, WeightArena, _f32


def get_motion_module(in_channels, motion_module_type: str, motion_module_kwargs: dict):
    if motion_module_type == "Vanilla":
        return VanillaTemporalModule(in_channels=in_channels, **motion_module_kwargs)
    raise ValueError(motion_module_type)


class PositionalEncoding(nn.Module):
    def __init__(self, d_model, dropout=0.0, max_len=24):
        super().__init__()
        position = torch.arange(max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(1, max_len, d_model)
        pe[0, :, 0::2] = torch.sin(position * div_term)
        pe[0, :, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe)  # in the checkpoint: ...pos_encoder.pe
        self.max_len = max_len
        self.table = None

    def pack(self, arena: WeightArena, dtype):
        self.table = arena.add((self.pe.shape[1], self.pe.shape[2]), torch.float32, lambda: _f32(self.pe)[0])


class VersatileAttention(Attention):
    def __init__(self, attention_mode=None, cross_frame_attention_mode=None, temporal_position_encoding=False,
                 temporal_position_encoding_max_len=24, *args, **kwargs):
        super().__init__(*args, **kwargs)
        assert attention_mode == "Temporal"
        self.attention_mode = attention_mode
        self.is_cross_attention = kwargs.get("cross_attention_dim") is not None
        if self.is_cross_attention:
            raise NotImplementedError("Temporal_Cross is not used by inference-v{1,2}.yaml")
        self.pos_encoder = PositionalEncoding(kwargs["query_dim"], max_len=temporal_position_encoding_max_len) \
            if temporal_position_encoding else None

    def pack(self, arena, dtype, fold_ln=None):
        pe = self.pos_encoder.pe if (fold_ln is not None and self.pos_encoder is not None) else None
        super().pack(arena, dtype, fold_ln=fold_ln, pe=pe)
        if self.pos_encoder is not None:
            self.pos_encoder.pack(arena, dtype)
        # operands of the one-launch form (K.tattn_fused: C = 320, 8 heads, 16 frames): the UNFOLDED q|k|v weights in fragment
        # order, the LayerNorm weight and (LayerNorm bias + positional encoding) per frame -- the kernel normalises its x tile in place
        self.tfrag = self.tgamma = self.tbias_pe = None
        # (the kernel has no projection-bias operand: with attention_bias=True -- no reference config -- the two-launch path runs)
        no_bias = all(getattr(m, "bias", None) is None for m in (self.to_q, self.to_k, self.to_v))
        if self.fold is not None and pe is not None and no_bias and (self.heads, self.inner_dim, self.query_dim) == (8, 320, 320):
            from .layers import _f32, frag_order_tattn
            self.tfrag = arena.add((368640,), dtype, lambda: frag_order_tattn(torch.cat([_f32(m.weight) for m in (self.to_q, self.to_k, self.to_v)], 0)))
            self.tgamma = arena.add((320,), torch.float32, lambda: _f32(fold_ln.weight))
            self.tbias_pe = arena.add((pe.shape[-2], 320), torch.float32,
                                      lambda: _f32(pe).reshape(pe.shape[-2], 320).to(fold_ln.bias.device) + _f32(fold_ln.bias)[None, :])
            self.tln_eps = fold_ln.eps

    def pos_table(self, frames: int):
        if self.pos_encoder is None:
            return None
        if frames > self.pos_encoder.max_len:
            raise ValueError(f"video_length {frames} exceeds temporal_position_encoding_max_len {self.pos_encoder.max_len}")
        return self.pos_encoder.table.t[:frames]


class TemporalTransformerBlock(nn.Module):
    def __init__(self, dim, num_attention_heads, attention_head_dim, attention_block_types=("Temporal_Self", "Temporal_Self"),
                 dropout=0.0, cross_attention_dim=768, activation_fn="geglu", attention_bias=False, upcast_attention=False,
                 cross_frame_attention_mode=None, temporal_position_encoding=False, temporal_position_encoding_max_len=24, **_):
        super().__init__()
        self.attention_blocks = nn.ModuleList([
            VersatileAttention(attention_mode=name.split("_")[0],
                               cross_attention_dim=cross_attention_dim if name.endswith("_Cross") else None,
                               query_dim=dim, heads=num_attention_heads, dim_head=attention_head_dim, bias=attention_bias,
                               temporal_position_encoding=temporal_position_encoding,
                               temporal_position_encoding_max_len=temporal_position_encoding_max_len)
            for name in attention_block_types])
        self.norms = nn.ModuleList([HipLayerNorm(dim) for _ in attention_block_types])
        self.ff = FeedForward(dim, activation_fn=activation_fn)
        self.ff_norm = HipLayerNorm(dim)

    def pack(self, arena, dtype):
        # LayerNorm (+ positional encoding) folded into the q|k|v projection, ff_norm into the GEGLU projection:
        # (LN(x) + pe) W^T = LN(x) W^T + pe W^T, the second term is a per-frame row bias (LnFold.pe)
        for attn, norm in zip(self.attention_blocks, self.norms):
            attn.pack(arena, dtype, fold_ln=norm)
            norm.pack(arena, dtype)
        self.ff.pack(arena, dtype, fold_ln=self.ff_norm)
        self.ff_norm.pack(arena, dtype)

    def forward(self, x: torch.Tensor, ctx: ExecCtx, tokens: int, proj_out=None) -> torch.Tensor:
        """x: [(b f n), C] rows.  proj_out = (fragment-ordered weight, bias, residual rows) of the temporal transformer this block ends:
        applied in the feed-forward's launch where possible (ca_ff_fused, ABI v12) -- the result then carries `_proj_out_done`."""
        rows, C = x.shape
        nblk = len(self.attention_blocks)
        for i, (attn, norm) in enumerate(zip(self.attention_blocks, self.norms)):
            # the next folded LayerNorm (the following attention's, or the feed-forward's) takes its statistics from this
            # projection's epilogue where the library can (K.gemm(row_sums=True))
            nxt = (self.attention_blocks[i + 1].fold if i + 1 < nblk else self.ff.fold) is not None
            if attn.fold is not None:
                o = attn(x.view(1, rows, C), residual=x.view(1, rows, C), temporal=(ctx.b, ctx.f, tokens),
                         ln=(K.RowStats(x, norm.eps, K.row_sums_of(x)), attn.fold), row_sums=nxt)
            else:
                n = norm.run(x, pos=attn.pos_table(ctx.f), rows_per_frame=tokens, frames=ctx.f)
                o = attn(n.view(1, rows, C), residual=x.view(1, rows, C), temporal=(ctx.b, ctx.f, tokens), row_sums=nxt)
            x = K.carry_row_sums(o.view(rows, C), o)
        if self.ff.fold is None:
            return self.ff.run(self.ff_norm.run(x), residual=x)
        if proj_out is not None and K.row_sums_of(x) is None:
            y = self.ff.run_with_proj_out(x, x, *proj_out)
            if y is not None:
                y._proj_out_done = True
                return y
        return self.ff.run(x, residual=x, sums=K.row_sums_of(x))


class TemporalTransformer3DModel(nn.Module):
    def __init__(self, in_channels, num_attention_heads, attention_head_dim, num_layers, norm_num_groups=32, **kw):
        super().__init__()
        inner = num_attention_heads * attention_head_dim
        self.norm = HipGroupNorm(norm_num_groups, in_channels, eps=1e-6)
        self.proj_in = HipLinear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([
            TemporalTransformerBlock(dim=inner, num_attention_heads=num_attention_heads,
                                     attention_head_dim=attention_head_dim, **kw) for _ in range(num_layers)])
        self.proj_out = HipLinear(inner, in_channels)

    def pack(self, arena, dtype):
        self.norm.pack(arena, dtype)
        self.proj_in.pack(arena, dtype)
        for b in self.transformer_blocks:
            b.pack(arena, dtype)
        self.proj_out.pack(arena, dtype)
        # proj_out once more in the fragment order of the output stage behind the last block's feed-forward (ca_ff_fused, ABI v12)
        self.proj_out_frag = None
        po = self.proj_out
        if tuple(po.weight.shape) == (320, 320) and getattr(self.transformer_blocks[-1].ff, "w2f", None) is not None:
            from .layers import frag_order_wout
            self.proj_out_frag = arena.add((102400,), dtype, lambda: frag_order_wout(_f32(po.weight)))

    def forward(self, x: torch.Tensor, ctx: ExecCtx) -> torch.Tensor:
        images, h, w, c = x.shape
        rows = images * h * w
        y = self.norm.run(x)  # per image ('(b f) c h w', motion_module.py:139-144)
        y = self.proj_in.run(y.view(rows, c), row_sums=self.transformer_blocks[0].attention_blocks[0].fold is not None)
        pfrag = getattr(self, "proj_out_frag", None)
        last = len(self.transformer_blocks) - 1
        for k, blk in enumerate(self.transformer_blocks):
            y = blk(y, ctx, h * w, proj_out=(pfrag, self.proj_out.b, x.view(rows, c)) if (k == last and pfrag is not None) else None)
        if getattr(y, "_proj_out_done", False):
            return y.view(images, h, w, c)
        return self.proj_out.run(y, residual=x.view(rows, c)).view(images, h, w, c)


class VanillaTemporalModule(nn.Module):
    def __init__(self, in_channels, num_attention_heads=8, num_transformer_block=2,
                 attention_block_types=("Temporal_Self", "Temporal_Self"), cross_frame_attention_mode=None,
                 temporal_position_encoding=False, temporal_position_encoding_max_len=24, temporal_attention_dim_div=1,
                 zero_initialize=True):
        super().__init__()
        self.temporal_transformer = TemporalTransformer3DModel(
            in_channels=in_channels, num_attention_heads=num_attention_heads,
            attention_head_dim=in_channels // num_attention_heads // temporal_attention_dim_div,
            num_layers=num_transformer_block, attention_block_types=attention_block_types,
            cross_frame_attention_mode=cross_frame_attention_mode, temporal_position_encoding=temporal_position_encoding,
            temporal_position_encoding_max_len=temporal_position_encoding_max_len)
        if zero_initialize:  # motion_module.py:76-77; real checkpoints overwrite it
            nn.init.zeros_(self.temporal_transformer.proj_out.weight)
            nn.init.zeros_(self.temporal_transformer.proj_out.bias)

    def pack(self, arena, dtype):
        self.temporal_transformer.pack(arena, dtype)

    def forward(self, x: torch.Tensor, ctx: ExecCtx) -> torch.Tensor:
        return self.temporal_transformer(x, ctx)
