"""Leaf layers of the HIP execution path: parameter containers with the reference's checkpoint
names + packed, kernel-friendly device copies of their weights.

Master parameters (nn.Parameter, any float dtype, any device) exist for state-dict
compatibility with the reference's checkpoints.  `WeightArena` gathers every packed
tensor of a model into ONE contiguous device buffer (so multi-GPU start-up is a single RCCL
broadcast) laid out as the kernels want it:
  * Linear / 1x1 conv:  [N, K] row-major in the activation dtype (both MFMA operands K-contiguous)
  * 3x3 conv:           [Cout, kh, kw, Cin]  (Cin padded to a multiple of 8 with zeros)
  * fused projections:  q|k|v (self-attention) and k|v (cross-attention) concatenated along N
  * GEGLU:              rows interleaved (h0, g0, h1, g1, ...) so the gate sits next to its value
  * biases, norm affine parameters, positional tables: fp32
  * K = 320 projections with N >= 960 (q|k|v, GEGLU of the 64x64-latent level): a second copy in MFMA-fragment order
    (`frag_order`, = ca_pack_w_frag) for the activation-resident kernel, whose waves read W fragments straight from L2
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
from torch import nn

from . import kernels as K
from .context import dispatch

ALIGN = 256  # bytes


class Packed:
    """Handle to one packed tensor inside the arena (valid after WeightArena.finalize)."""

    __slots__ = ("shape", "dtype", "fill", "offset", "t", "frag")

    def __init__(self, shape, dtype, fill):
        self.shape = tuple(int(s) for s in shape)
        self.dtype = dtype
        self.fill = fill
        self.offset = -1
        self.t: Optional[torch.Tensor] = None
        self.frag = None  # (Packed, geglu): the fragment-ordered twin of this weight (ca_gemm_args.w_frag), linked in finalize

    @property
    def nbytes(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n * torch.empty((), dtype=self.dtype).element_size()


class WeightArena:
    def __init__(self):
        self.items: List[Packed] = []
        self.buffer: Optional[torch.Tensor] = None

    def add(self, shape, dtype, fill: Callable[[], torch.Tensor]) -> Packed:
        p = Packed(shape, dtype, fill)
        self.items.append(p)
        return p

    def finalize(self, device) -> torch.Tensor:
        off = 0
        for p in self.items:
            p.offset = off
            off += (p.nbytes + ALIGN - 1) // ALIGN * ALIGN
        self.buffer = torch.zeros(max(off, ALIGN), dtype=torch.uint8, device=device)
        for p in self.items:
            view = self.buffer[p.offset:p.offset + p.nbytes].view(p.dtype).view(p.shape)
            src = p.fill()
            assert tuple(src.shape) == p.shape, (tuple(src.shape), p.shape)
            view.copy_(src.to(device=device, dtype=p.dtype, non_blocking=False))
            p.t = view
            p.fill = None
        for p in self.items:
            if p.frag is not None:  # kernels.gemm hands the twin over with the weight
                p.t._frag = (p.frag[0].t, bool(p.frag[1]))
        return self.buffer

    @property
    def nbytes(self) -> int:
        return 0 if self.buffer is None else self.buffer.numel()


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.detach().float()


def frag_order(w: torch.Tensor, geglu: bool) -> torch.Tensor:
    """[N, 320] -> the same elements in the order ca_gemm_args.w_frag takes (= ca_pack_w_frag, csrc/ca_gemm_ar.h): 16-byte piece
    L (64 per MFMA tile) of tile j (4 per 64-column panel) of 32-deep chunk kq (10) of panel pn holds
    W[pn * 64 + col(j, L & 15)][kq * 32 + (L >> 4) * 8 : + 8], col = the weight-row interleave behind the 16-byte stores."""
    n, k = w.shape
    assert k == 320 and n % 64 == 0
    j = torch.arange(4).view(4, 1)
    i = torch.arange(16).view(1, 16)
    col = (16 * (i >> 2) + 4 * j + (i & 3)) if geglu else (32 * (j >> 1) + 8 * (i >> 2) + 4 * (j & 1) + (i & 3))  # [j, i]
    rows = (torch.arange(n // 64).view(-1, 1, 1) * 64 + col.view(1, 4, 16)).to(w.device)                           # [pn, j, i]
    x = w[rows.reshape(-1)].view(n // 64, 4, 16, 10, 4, 8)  # [pn, j, i = L & 15, kq, g = L >> 4, e]
    return x.permute(0, 3, 1, 4, 2, 5).reshape(n, k).contiguous()  # [pn, kq, j, g, i, e]


def frag_order2(w: torch.Tensor) -> torch.Tensor:
    """[320, 1280] (the feed-forward's output projection at C = 320) -> the order ca_ff_args.w2_frag takes (= ca_pack_w2_frag,
    csrc/ca_ff_fused.h): 16-byte piece L of MFMA tile j (5 per 80-column range of a consumer wave) of wave wc of h chunk pn
    (40 chunks of 32) holds W[wc * 80 + col5(j, L & 15)][pn * 32 + (L >> 4) * 8 : + 8]."""
    assert tuple(w.shape) == (320, 1280)
    j = torch.arange(5).view(5, 1)
    i = torch.arange(16).view(1, 16)
    col = torch.where(j == 4, 64 + i, 32 * (j >> 1) + 8 * (i >> 2) + 4 * (j & 1) + (i & 3))  # [j, i]
    rows = (torch.arange(4).view(4, 1, 1) * 80 + col.view(1, 5, 16)).to(w.device)                # [wc, j, i]
    x = w[rows.reshape(-1)].view(4, 5, 16, 40, 4, 8)  # [wc, j, i = L & 15, pn, g = L >> 4, e]
    return x.permute(3, 0, 1, 4, 2, 5).reshape(320, 1280).contiguous()  # [pn, wc, j, g, i, e]


def frag_order_tattn(wqkv: torch.Tensor) -> torch.Tensor:
    """[960, 320] (rows Wq | Wk | Wv of 8 heads x 40) -> the flat order ca_tattn_args.w_frag takes (= ca_pack_w_tattn,
    csrc/ca_tattn_fused.h): 16-byte piece L of column tile j of 32-deep chunk kq of pass ps (q, k, v) of head wv + 4 hi holds
    W_ps[head * 40 + 16 j + (L & 15)][kq * 32 + (L >> 4) * 8 : + 8], zeros where 16 j + (L & 15) >= 40."""
    assert tuple(wqkv.shape) == (960, 320)
    wpad = wqkv.new_zeros(3, 8, 48, 320)
    wpad[:, :, :40] = wqkv.view(3, 8, 40, 320)
    x = wpad.view(3, 2, 4, 3, 16, 10, 4, 8)  # [ps, hi, wv, j, i = L & 15, kq, g = L >> 4, e]
    return x.permute(2, 1, 0, 5, 3, 6, 4, 7).reshape(-1).contiguous()  # [wv, hi, ps, kq, j, g, i, e]


def frag_order_xattn(wq: torch.Tensor) -> torch.Tensor:
    """[320, 320] (Wq of 8 heads x 40, LayerNorm-folded) -> the flat order ca_xattn_args.wq_frag takes (= ca_xattn_pack_w,
    csrc/ca_xattn_fused.h): 16-byte piece L of column tile j of 32-deep chunk kq of head wv + 4 hi holds
    Wq[head * 40 + 16 j + (L & 15)][kq * 32 + (L >> 4) * 8 : + 8], zeros where 16 j + (L & 15) >= 40."""
    assert tuple(wq.shape) == (320, 320)
    wpad = wq.new_zeros(8, 48, 320)
    wpad[:, :40] = wq.view(8, 40, 320)
    x = wpad.view(2, 4, 3, 16, 10, 4, 8)  # [hi, wv, j, i = L & 15, kq, g = L >> 4, e]
    return x.permute(1, 0, 4, 2, 5, 3, 6).reshape(-1).contiguous()  # [wv, hi, kq, j, g, i, e]


def frag_order_wout(w: torch.Tensor) -> torch.Tensor:
    """[320, 320] (an attention's to_out[0].weight at C = 320) -> the flat order ca_tattn_args.w_out_frag / ca_xattn_args.w_out_frag take
    (= ca_pack_w_out, csrc/ca_attn_out.h): 16-byte piece L of column tile j (5 per 80-column group) of 32-deep chunk kq of column group
    cg holds W[cg * 80 + col5(j, L & 15)][kq * 32 + (L >> 4) * 8 : + 8]."""
    assert tuple(w.shape) == (320, 320)
    j = torch.arange(5).view(5, 1)
    i = torch.arange(16).view(1, 16)
    col = torch.where(j == 4, 64 + i, 32 * (j >> 1) + 8 * (i >> 2) + 4 * (j & 1) + (i & 3))  # [j, i]
    rows = (torch.arange(4).view(4, 1, 1) * 80 + col.view(1, 5, 16)).to(w.device)                # [cg, j, i]
    x = w[rows.reshape(-1)].view(4, 5, 16, 10, 4, 8)  # [cg, j, i = L & 15, kq, g = L >> 4, e]
    return x.permute(0, 3, 1, 4, 2, 5).reshape(-1).contiguous()  # [cg, kq, j, g, i, e]


def frag_wanted(n: int, k: int) -> bool:
    """The shapes the activation-resident kernel takes (ca_gemm.hip ar_eligible): K = 320, N a multiple of 320, N >= 960."""
    return k == 320 and n % 320 == 0 and n >= 960


class HipLinear(nn.Module):
    """nn.Linear-compatible parameters (weight [N,K], bias [N]); executes via ca_gemm."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        nn.init.normal_(self.weight, std=in_features ** -0.5)
        if bias:
            nn.init.zeros_(self.bias)
        self.w: Optional[Packed] = None
        self.b: Optional[Packed] = None

    def pack(self, arena: WeightArena, dtype):
        self.w = arena.add((self.out_features, self.in_features), dtype, lambda: _f32(self.weight))
        self.b = arena.add((self.out_features,), torch.float32, lambda: _f32(self.bias)) if self.bias is not None else None

    def run(self, a: torch.Tensor, **kw) -> torch.Tensor:
        return K.gemm(a, self.w.t, bias=None if self.b is None else self.b.t, **kw)


class HipConv1x1(nn.Module):
    """Conv2d 1x1 parameters (weight [Cout,Cin,1,1]); a plain row-major GEMM in NHWC."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, 1, 1))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        nn.init.normal_(self.weight, std=in_channels ** -0.5)
        self.w = self.b = None

    def pack(self, arena: WeightArena, dtype):
        self.w = arena.add((self.out_channels, self.in_channels), dtype, lambda: _f32(self.weight).reshape(self.out_channels, self.in_channels))
        self.b = arena.add((self.out_channels,), torch.float32, lambda: _f32(self.bias))

    def run(self, a: torch.Tensor, **kw) -> torch.Tensor:
        return K.gemm(a, self.w.t, bias=self.b.t, **kw)


class HipConv3x3(nn.Module):
    """Conv2d 3x3 pad 1 parameters (weight [Cout,Cin,3,3]); NHWC implicit GEMM via ca_conv3x3."""

    def __init__(self, in_channels: int, out_channels: int, stride: int = 1):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride
        self.cin_pad = (in_channels + 7) // 8 * 8
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, 3, 3))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        nn.init.normal_(self.weight, std=(9 * in_channels) ** -0.5)
        self.w = self.b = None
        # True: pack() also stores the Winograd form U of a deep stride-1 convolution.  Which CALLS take the route is the library's
        # decision per shape (wino_workspace_bytes in ca_gemm.hip: >= 1280 input channels, or 640 .. 1279 at <= 4096 tiles -- 8-frame windows,
        # the de-duplicated ControlNet batch -- nearest-x2 upsampling included, asymmetric padding and stride 2 not); a model that never
        # reaches such a shape can set this False before prepare() and save 16/9 of the weight per convolution (VAE: never packed, < 640)
        self.winograd = True

    def _packed_weight(self) -> torch.Tensor:
        w = _f32(self.weight).permute(0, 2, 3, 1)  # [Cout, kh, kw, Cin]
        if self.cin_pad != self.in_channels:
            w = torch.nn.functional.pad(w, (0, self.cin_pad - self.in_channels))
        return w.contiguous()

    def _winograd_weight(self) -> torch.Tensor:
        """U [16, Cout, Cin] = G g G^T per (Cout, Cin) from the fp32 weights (one rounding to the activation type): F(2x2, 3x3),
        csrc/ca_conv_wino.h."""
        g = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float32, device=self.weight.device)
        u = torch.einsum("xk,oikl,yl->xyoi", g, _f32(self.weight), g)  # weight [Cout, Cin, kh, kw]
        return u.reshape(16, self.out_channels, self.in_channels).contiguous()

    def pack(self, arena: WeightArena, dtype):
        self.w = arena.add((self.out_channels, 3, 3, self.cin_pad), dtype, self._packed_weight)
        self.b = arena.add((self.out_channels,), torch.float32, lambda: _f32(self.bias))
        # the deep stride-1 convolutions also in Winograd form: ca_conv3x3 takes it at the small-latent levels (ca_conv_args.w_wino, ABI v12)
        self.u = None
        if (dispatch.conv_winograd and self.stride == 1 and self.winograd and self.in_channels >= 640 and self.in_channels % 64 == 0
                and self.out_channels % 320 == 0):
            self.u = arena.add((16, self.out_channels, self.in_channels), dtype, self._winograd_weight)

    def run(self, x: torch.Tensor, **kw) -> torch.Tensor:
        u = getattr(self, "u", None)
        return K.conv3x3(x, self.w.t, bias=self.b.t, stride=self.stride, w_wino=None if u is None else u.t, **kw)


class HipGroupNorm(nn.Module):
    def __init__(self, num_groups: int, num_channels: int, eps: float = 1e-5):
        super().__init__()
        self.num_groups, self.num_channels, self.eps = num_groups, num_channels, eps
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))
        self.g = self.b = None

    def pack(self, arena: WeightArena, dtype):
        self.g = arena.add((self.num_channels,), torch.float32, lambda: _f32(self.weight))
        self.b = arena.add((self.num_channels,), torch.float32, lambda: _f32(self.bias))

    def run(self, x: torch.Tensor, *, x2: Optional[torch.Tensor] = None, frames_per_stat: int = 1, act: int = K.ACT_NONE):
        return K.group_norm(x, self.g.t, self.b.t, x2=x2, groups=self.num_groups, frames_per_stat=frames_per_stat,
                            eps=self.eps, act=act)


class HipLayerNorm(nn.Module):
    def __init__(self, dim: int, eps: float = 1e-5):
        super().__init__()
        self.dim, self.eps = dim, eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))
        self.g = self.b = None

    def pack(self, arena: WeightArena, dtype):
        self.g = arena.add((self.dim,), torch.float32, lambda: _f32(self.weight))
        self.b = arena.add((self.dim,), torch.float32, lambda: _f32(self.bias))

    def run(self, x: torch.Tensor, *, pos: Optional[torch.Tensor] = None, rows_per_frame: int = 1, frames: int = 1):
        return K.layer_norm(x, self.g.t, self.b.t, pos=pos, rows_per_frame=rows_per_frame, frames=frames, eps=self.eps)


def pack_concat_rows(arena: WeightArena, dtype, mods: Sequence[nn.Module]) -> Packed:
    """One [sum N_i, K] matrix from several Linear weights (fused q|k|v or k|v projections)."""
    n = sum(m.weight.shape[0] for m in mods)
    k = mods[0].weight.shape[1]
    return arena.add((n, k), dtype, lambda: torch.cat([_f32(m.weight).reshape(m.weight.shape[0], -1) for m in mods], 0))


def pack_concat_bias(arena: WeightArena, mods: Sequence[nn.Module]) -> Packed:
    n = sum(m.weight.shape[0] for m in mods)
    return arena.add((n,), torch.float32, lambda: torch.cat([_f32(m.bias) for m in mods], 0))


def ln_fold_enabled() -> bool:
    """context.dispatch.ln_fold = False keeps LayerNorm as its own kernel (A/B measurements, set from outside the package)."""
    from .context import dispatch
    return bool(dispatch.ln_fold)


class LnFold:
    """Packed operands of `LayerNorm -> Linear(s)` run as ONE ca_gemm on the un-normalised activations
    (ca_gemm_args.ln_stats): W' = cat(W_i) diag(gamma) [rows GEGLU-interleaved if asked], colsum(W') of the
    ROUNDED W', bias' = cat(W_i) beta + cat(b_i).  With a positional table `pe` [max_len, K] (temporal
    attention: (LN(x) + pe) W^T) also the per-frame row bias pe W^T [max_len, N]."""

    def __init__(self, arena: WeightArena, dtype, ln: "HipLayerNorm", mods: Sequence[nn.Module], geglu: bool = False, pe=None):
        n = sum(m.weight.shape[0] for m in mods)
        k = mods[0].weight.shape[1]
        self.eps = ln.eps

        def w_cat():
            return torch.cat([_f32(m.weight).reshape(m.weight.shape[0], -1) for m in mods], 0)

        def w_fold():
            w = w_cat() * _f32(ln.weight)[None, :]
            return geglu_interleave(w) if geglu else w

        def bias():
            b = w_cat() @ _f32(ln.bias)
            b = b + torch.cat([_f32(m.bias) if getattr(m, "bias", None) is not None else torch.zeros(m.weight.shape[0], device=m.weight.device) for m in mods], 0)
            return geglu_interleave(b) if geglu else b

        self.w = arena.add((n, k), dtype, w_fold)
        self.w_fold = w_fold  # (the fp32 folded weight: other packings of it -- attention_processor.Attention.pack)
        if frag_wanted(n, k):
            self.w.frag = (arena.add((n, k), dtype, lambda: frag_order(w_fold(), geglu)), geglu)
        self.cs = arena.add((n,), torch.float32, lambda: w_fold().to(dtype).float().sum(1))
        self.b = arena.add((n,), torch.float32, bias)
        self.pe = None
        if pe is not None:
            self.pe = arena.add((pe.shape[-2], n), torch.float32, lambda: _f32(pe).reshape(pe.shape[-2], k).to(mods[0].weight.device) @ w_cat().to(dtype).float().t())
        self._rb_cache = {}

    def rowbias(self, b: int, f: int) -> torch.Tensor:
        """[b*f, N] fp32: row group g = (batch, frame) gets pe[frame] W^T (rows are in (b f n) order)."""
        if f > self.pe.t.shape[0]:
            raise ValueError(f"video_length {f} exceeds temporal_position_encoding_max_len {self.pe.t.shape[0]}")
        key = (b, f, self.pe.t.data_ptr())
        rb = self._rb_cache.get(key)
        if rb is None:
            rb = self._rb_cache[key] = self.pe.t[:f].repeat(b, 1).contiguous()
        return rb


def geglu_interleave(w: torch.Tensor) -> torch.Tensor:
    """[2D, ...] (value rows then gate rows) -> rows (v0, g0, v1, g1, ...)."""
    d = w.shape[0] // 2
    return torch.stack([w[:d], w[d:]], dim=1).reshape(w.shape)
