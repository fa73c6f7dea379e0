"""Thin torch-tensor front end of the C ABI (device pointers + current HIP stream).

torch is used only for device memory and streams; every arithmetic op below is a call into
`libcontrolanimate_hip.so`.  All activations are channels-last ("NHWC"): `[images, H, W, C]`.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _capi
from .context import dispatch
from ._capi import (AttnArgs, ConvArgs, FfArgs, TattnArgs, XattnArgs, GemmArgs, GroupNormArgs, LayerNormArgs, CA_ACT_NONE,
                    CA_ACT_SILU, CA_BF16, CA_F16, check, lib)

ACT_NONE, ACT_SILU = CA_ACT_NONE, CA_ACT_SILU
ACT_QUICK_GELU, ACT_GELU = 2, 3  # CA_ACT_QUICK_GELU / CA_ACT_GELU (CLIP MLPs)


def dt_code(dtype: torch.dtype) -> int:
    if dtype == torch.bfloat16:
        return CA_BF16
    if dtype == torch.float16:
        return CA_F16
    raise TypeError(f"activations must be bf16 or fp16, got {dtype}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _req_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _capi.CAHipError("HIP kernels need device tensors (no CPU fallback)")


_plan_sink = None  # a list: receives the kernel label (ca_gemm_plan_name / ca_conv3x3_plan_name) of every launch -- bench.py, tests


def _record_plan(query, args) -> None:
    if _plan_sink is not None:
        buf = C.create_string_buffer(64)
        _plan_sink.append(buf.value.decode() if query(C.byref(args), buf, 64) == 0 else "?")


def gemm(a: torch.Tensor, w: torch.Tensor, *, a2: Optional[torch.Tensor] = None,
         bias: Optional[torch.Tensor] = None, rowbias: Optional[torch.Tensor] = None,
         rows_per_group: int = 0, residual: Optional[torch.Tensor] = None, alpha: float = 1.0,
         post_scale: float = 1.0, act: int = ACT_NONE, geglu: bool = False, out_f32: bool = False,
         out: Optional[torch.Tensor] = None, ln=None, row_sums: bool = False) -> torch.Tensor:
    """out[M, N] = epilogue(cat(a, a2)[M, K] @ w[N, K]^T); see ca_gemm in the header.
    ln = (row_stats(a), colsum(w) fp32 [N]): LayerNorm of `a` folded in (w, bias packed accordingly)."""
    _req_cuda(a, w, a2, bias, rowbias, residual, out)
    assert a.dim() == 2 and a.stride(1) == 1 and w.dim() == 2 and w.is_contiguous()
    m, k1 = a.shape
    k2 = 0
    if a2 is not None:
        assert a2.dim() == 2 and a2.stride(1) == 1 and a2.shape[0] == m and a2.dtype == a.dtype
        k2 = a2.shape[1]
    n = w.shape[0]
    assert w.shape[1] == k1 + k2 and w.dtype == a.dtype
    ncols = n // 2 if geglu else n
    if out is None:
        out = torch.empty((m, ncols), device=a.device, dtype=torch.float32 if out_f32 else a.dtype)
    assert out.shape == (m, ncols) and out.stride(1) == 1
    if residual is not None:
        assert residual.shape == (m, n) and residual.stride(1) == 1 and residual.dtype == a.dtype
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == n
    if rowbias is not None:
        assert rowbias.dtype == torch.float32 and rowbias.dim() == 2 and rowbias.shape[1] == n and rows_per_group > 0
    args = GemmArgs(a=_p(a), a2=_p(a2), w=_p(w), c=_p(out), bias=_p(bias), rowbias=_p(rowbias),
                    residual=_p(residual), lda=a.stride(0), lda2=a2.stride(0) if a2 is not None else 0,
                    ldc=out.stride(0), ld_res=residual.stride(0) if residual is not None else 0,
                    ld_rowbias=rowbias.stride(0) if rowbias is not None else 0,
                    m=m, n=n, k1=k1, k2=k2, rows_per_group=rows_per_group, alpha=alpha,
                    post_scale=post_scale, act=act, geglu=int(geglu), out_f32=int(out_f32),
                    dtype=dt_code(a.dtype))
    frag = getattr(w, "_frag", None)  # (tensor, geglu flag it was packed for): attach_w_frag
    if frag is not None and frag[1] == bool(geglu) and dispatch.gemm_ar:
        args.w_frag = _p(frag[0])
    if ln is not None:
        st, cs = ln
        assert cs.dtype == torch.float32 and cs.numel() == n
        args.ln_colsum = _p(cs)
        if isinstance(st, RowStats):  # statistics on demand: inside the GEMM where the library can, else the partial
            # sums the producing GEMM left (row_sums=True), else the separate pass
            assert st.x.data_ptr() == a.data_ptr() and st.x.shape == a.shape, "RowStats belongs to another tensor"
            args.ln_eps = st.eps
            if lib().ca_gemm_ln_inline_supported(C.byref(args)):
                pass
            elif st.sums is not None:
                rs, parts = st.sums
                assert rs.shape == (m, parts, 2) and rs.dtype == torch.float32 and rs.is_contiguous()
                args.ln_stats, args.ln_parts = _p(rs), min(parts, 4)
                # (more than 4 partial sums per row -- a producer on the 256 x 320 kernel leaves one per 80-column wave quarter --
                #  are always finished first: the consuming epilogues add at most 4)
                if parts > 4 or lib().ca_gemm_wants_finished_stats(C.byref(args)):
                    # the kernel the plan prefers for this shape (256 x 320 tiles) reads finished (mean, rstd): a 3 us pass
                    # over [m, parts, 2] instead of the consumer's epilogue adding the parts (ABI v8)
                    fin = getattr(st, "_finished", None)
                    if fin is None:
                        fin = torch.empty((m, 2), device=a.device, dtype=torch.float32)
                        check(lib().ca_ln_finish_sums(_p(rs), parts, m, k1 + k2, st.eps, _p(fin), _stream()), "ca_ln_finish_sums")
                        st._finished = fin
                    args.ln_stats, args.ln_parts = _p(fin), 0
            else:
                st = st.tensor()
        if not isinstance(st, RowStats):
            assert st.dtype == torch.float32 and st.shape == (m, 2) and st.is_contiguous()
            args.ln_stats = _p(st)
    if hasattr(out, "_row_sums"):  # a caller-supplied `out` reused from an earlier call: its sums describe the old contents
        del out._row_sums
    if row_sums and dispatch.ln_row_sums and not geglu and not out_f32:
        parts = int(lib().ca_gemm_row_sums_parts(C.byref(args)))
        if parts > 0:  # the epilogue leaves (sum, sum of squares) per row and 320-column tile: the next LayerNorm's statistics
            rs = torch.empty((m, parts, 2), device=a.device, dtype=torch.float32)
            args.row_sums_out = _p(rs)
            out._row_sums = (rs, parts)
    wbytes = 0 if args.row_sums_out else int(lib().ca_gemm_workspace_bytes(C.byref(args)))
    if wbytes > 0:  # split-K slabs for the 8x8-latent level (allocator-cached, stream-ordered)
        ws = torch.empty((wbytes,), device=a.device, dtype=torch.uint8)
        args.workspace, args.workspace_bytes = _p(ws), wbytes
    _record_plan(lib().ca_gemm_plan_name, args)
    check(lib().ca_gemm(C.byref(args), _stream()), "ca_gemm")
    return out


def attach_w_frag(w: torch.Tensor, geglu: bool = False) -> torch.Tensor:
    """Gives a packed [N, 320] weight its fragment-ordered twin (ca_pack_w_frag, ABI v9): `gemm` hands both over and the K = 320
    projections of the 64x64-latent level run on the activation-resident kernel.  No-op for shapes that kernel cannot take.
    Called once per weight at prepare() time; the twin lives beside the arena (N x 640 bytes)."""
    _req_cuda(w)
    if w.dim() != 2 or w.shape[1] != 320 or w.shape[0] % 320 != 0 or not w.is_contiguous() or w.dtype not in (torch.float16, torch.bfloat16):
        return w
    dst = torch.empty_like(w)
    check(lib().ca_pack_w_frag(w.data_ptr(), w.shape[0], w.shape[1], int(bool(geglu)), dst.data_ptr(), _stream()), "ca_pack_w_frag")
    w._frag = (dst, bool(geglu))
    return w


def ff_fused(x: torch.Tensor, w1_frag: torch.Tensor, bias1: torch.Tensor, colsum1: torch.Tensor, w2_frag: torch.Tensor,
             bias2: Optional[torch.Tensor], ln_eps: float, residual: Optional[torch.Tensor] = None,
             ln_stats: Optional[torch.Tensor] = None, *, w_out_frag: Optional[torch.Tensor] = None, bias_out: Optional[torch.Tensor] = None,
             residual_out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """y = GEGLU(LN(x) W1^T + b1) W2^T + b2 + residual in one launch (ca_ff_fused, ABI v9: the 64x64-latent level's feed-forward,
    C = 320) -- or None where the library does not take the arguments (the caller then runs the two GEMMs).
    With w_out_frag (layers.frag_order_wout of the transformer's proj_out.weight; ABI v12) the launch also applies that projection
    and returns y Wout^T + bias_out + residual_out."""
    if not dispatch.ff_fused:
        return None
    _req_cuda(x, w1_frag, bias1, colsum1, w2_frag, bias2, residual, ln_stats, w_out_frag, bias_out, residual_out)
    if x.dim() != 2 or x.stride(1) != 1 or (residual is not None and (residual.shape != x.shape or residual.stride(1) != 1 or residual.dtype != x.dtype)):
        return None
    if w_out_frag is None and (bias_out is not None or residual_out is not None):
        return None
    if residual_out is not None and (residual_out.shape != x.shape or residual_out.stride(1) != 1 or residual_out.dtype != x.dtype):
        return None
    m, c = x.shape
    inner = w2_frag.shape[1]
    y = torch.empty((m, c), device=x.device, dtype=x.dtype)
    args = FfArgs(x=_p(x), w1_frag=_p(w1_frag), bias1=_p(bias1), colsum1=_p(colsum1), ln_stats=_p(ln_stats), w2_frag=_p(w2_frag),
                  bias2=_p(bias2), residual=_p(residual), y=_p(y), lda=x.stride(0), ldc=y.stride(0),
                  ld_res=residual.stride(0) if residual is not None else 0, m=m, c=c, inner=inner, ln_eps=float(ln_eps), dtype=dt_code(x.dtype),
                  w_out_frag=_p(w_out_frag), bias_out=_p(bias_out), residual_out=_p(residual_out),
                  ld_res_out=residual_out.stride(0) if residual_out is not None else 0)
    if not lib().ca_ff_fused_supported(C.byref(args)):
        return None
    if _plan_sink is not None:
        _plan_sink.append("ff_out128" if w_out_frag is not None else "ff_fused128")
    check(lib().ca_ff_fused(C.byref(args), _stream()), "ca_ff_fused")
    return y


def tattn_fused(x: torch.Tensor, w_frag: torch.Tensor, gamma: torch.Tensor, bias_pe: torch.Tensor, b: int, frames: int, tokens: int,
                heads: int, ln_eps: float, scale: float, *, w_out_frag: Optional[torch.Tensor] = None, bias_out: Optional[torch.Tensor] = None,
                residual: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """o = softmax(q k^T scale) v over the frame axis with q|k|v = (LayerNorm(x) + pe[frame]) Wqkv^T in one launch (ca_tattn_fused,
    ABI v10: the motion modules of the 64x64-latent level) -- or None where the library does not take the arguments (the caller
    then runs the folded q|k|v GEMM and attention_temporal).  x rows in (b f n) order; bias_pe [>= frames, C] fp32.
    With w_out_frag (layers.frag_order_wout of to_out[0].weight; ABI v12) the launch also applies the output projection and returns
    y = o Wout^T + bias_out + residual."""
    if not dispatch.tattn_fused:
        return None
    _req_cuda(x, w_frag, gamma, bias_pe, w_out_frag, bias_out, residual)
    if x.dim() != 2 or x.stride(1) != 1 or x.shape[0] != b * frames * tokens or bias_pe.shape[0] < frames or bias_pe.stride(1) != 1:
        return None
    if w_out_frag is None and (bias_out is not None or residual is not None):
        return None
    if residual is not None and (residual.shape != x.shape or residual.stride(1) != 1 or residual.dtype != x.dtype):
        return None
    c = x.shape[1]
    o = torch.empty((x.shape[0], c), device=x.device, dtype=x.dtype)
    args = TattnArgs(x=_p(x), w_frag=_p(w_frag), gamma=_p(gamma), bias_pe=_p(bias_pe), o=_p(o), lda=x.stride(0), ldo=o.stride(0),
                     ld_bias_pe=bias_pe.stride(0), batch=b, frames=frames, tokens=tokens, heads=heads, c=c, ln_eps=float(ln_eps),
                     scale=float(scale), dtype=dt_code(x.dtype), w_out_frag=_p(w_out_frag), bias_out=_p(bias_out), residual=_p(residual),
                     ld_res=residual.stride(0) if residual is not None else 0)
    if not lib().ca_tattn_fused_supported(C.byref(args)):
        return None
    if _plan_sink is not None:
        _plan_sink.append("tattn_out128" if w_out_frag is not None else "tattn_fused128")
    check(lib().ca_tattn_fused(C.byref(args), _stream()), "ca_tattn_fused")
    return o


def xattn_pack_kv(kv: torch.Tensor, kv_batches: int, rows_per_batch: int, nk: int, scale: float, row_offset: int = 0,
                  out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """The K | V rows of `kv` [kv_batches * rows_per_batch, 2 * 320] as the MFMA fragments ca_xattn_fused reads (ca_xattn_pack_kv,
    ABI v11; once per window and layer) -- or None for shapes the fused kernel does not take: 65..80 text keys, or (ABI v13) the 1..16
    image-prompt tokens of the IP-Adapter's second attention (row_offset = the first of them)."""
    if (not dispatch.xattn_fused or kv.dim() != 2 or kv.shape[1] != 640 or kv.stride(1) != 1 or not (64 < nk <= 80 or 1 <= nk <= 16)
            or kv.dtype not in (torch.float16, torch.bfloat16)):
        return None
    _req_cuda(kv)
    dst = out if out is not None else torch.empty((kv_batches, 8, _capi.XATTN_KV_FRAG_ELEMS), device=kv.device, dtype=kv.dtype)
    assert dst.shape == (kv_batches, 8, _capi.XATTN_KV_FRAG_ELEMS) and dst.dtype == kv.dtype and dst.is_contiguous()
    check(lib().ca_xattn_pack_kv(kv.data_ptr(), kv.stride(0), kv_batches, rows_per_batch, row_offset, nk, float(scale), dt_code(kv.dtype),
                                 dst.data_ptr(), _stream()), "ca_xattn_pack_kv")
    return dst


def xattn_fused(x: torch.Tensor, wq_frag: torch.Tensor, bias: Optional[torch.Tensor], kv_frag: torch.Tensor, images: int, tokens: int,
                frames_per_kv: int, kv_mod: int, nk: int, ln_eps: float, *, w_out_frag: Optional[torch.Tensor] = None,
                bias_out: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, kv_frag_ip: Optional[torch.Tensor] = None,
                nk_ip: int = 0, ip_scale: float = 1.0) -> Optional[torch.Tensor]:
    """o = softmax(q K^T scale) V with q = LayerNorm(x) Wq^T + bias in one launch (ca_xattn_fused, ABI v11: the text cross-attention of
    the 64x64-latent level) -- or None where the library does not take the arguments (the caller then runs the folded q GEMM and
    attention_cross).  kv_frag from xattn_pack_kv; image z uses its text batch (z // frames_per_kv) % kv_mod, kv_mod = 0 meaning `images`
    exactly as attention_cross does (the library refuses a launch that would index past the packed text batches).
    With w_out_frag (layers.frag_order_wout of to_out[0].weight; ABI v12) the launch also applies the output projection and returns
    y = o Wout^T + bias_out + residual.  With kv_frag_ip (xattn_pack_kv of the IP-Adapter's to_k_ip | to_v_ip projection of the nk_ip
    image-prompt tokens; ABI v13, needs w_out_frag) o = o_text + ip_scale * softmax(q K_ip^T scale) V_ip before that projection."""
    if not dispatch.xattn_fused or kv_frag is None:
        return None
    _req_cuda(x, wq_frag, bias, kv_frag, w_out_frag, bias_out, residual, kv_frag_ip)
    if kv_frag_ip is not None and (w_out_frag is None or kv_frag_ip.shape != kv_frag.shape or kv_frag_ip.dtype != x.dtype or not kv_frag_ip.is_contiguous()):
        return None
    if x.dim() != 2 or x.stride(1) != 1 or x.shape[0] != images * tokens:
        return None
    if w_out_frag is None and (bias_out is not None or residual is not None):
        return None
    if residual is not None and (residual.shape != x.shape or residual.stride(1) != 1 or residual.dtype != x.dtype):
        return None
    assert kv_frag.dtype == x.dtype and wq_frag.dtype == x.dtype, "ca_xattn_fused: x, the packed Wq and the packed K/V must share one dtype"
    c = x.shape[1]
    o = torch.empty((x.shape[0], c), device=x.device, dtype=x.dtype)
    args = XattnArgs(x=_p(x), wq_frag=_p(wq_frag), bias=_p(bias), kv_frag=_p(kv_frag), o=_p(o), lda=x.stride(0), ldo=o.stride(0), m=x.shape[0],
                     tokens=tokens, frames_per_kv=frames_per_kv, kv_mod=kv_mod if kv_mod > 0 else images, kv_batches=kv_frag.shape[0],
                     nk=nk, heads=8, c=c, ln_eps=float(ln_eps), dtype=dt_code(x.dtype), w_out_frag=_p(w_out_frag), bias_out=_p(bias_out),
                     residual=_p(residual), ld_res=residual.stride(0) if residual is not None else 0, kv_frag_ip=_p(kv_frag_ip),
                     nk_ip=nk_ip if kv_frag_ip is not None else 0, ip_scale=float(ip_scale) if kv_frag_ip is not None else 0.0)
    if not lib().ca_xattn_fused_supported(C.byref(args)):
        return None
    if _plan_sink is not None:
        _plan_sink.append(("xattn_ip_out128" if kv_frag_ip is not None else "xattn_out128") if w_out_frag is not None else "xattn_fused128")
    check(lib().ca_xattn_fused(C.byref(args), _stream()), "ca_xattn_fused")
    return o


def row_sums_of(t: torch.Tensor):
    """The partial row sums the GEMM that produced `t` left for the next LayerNorm, or None."""
    return getattr(t, "_row_sums", None)


def carry_row_sums(dst: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
    """`dst` is a view of `src` (same rows): keep the producer's row sums attached."""
    s = getattr(src, "_row_sums", None)
    if s is not None:
        dst._row_sums = s
    return dst


class RowStats:
    """LayerNorm statistics of the rows of `x`, not computed yet: `gemm(x, w, ln=(RowStats(x, eps), colsum))` lets the GEMM
    compute them itself while it streams x (ca_gemm_args.ln_eps, ABI v6: the K = 320 weight-resident kernel) and falls
    back to the separate pass (`row_stats`) where the library cannot."""

    def __init__(self, x: torch.Tensor, eps: float = 1e-5, sums=None):
        # sums = (partial sums [rows, parts, 2], parts) left by the GEMM that produced x (gemm(..., row_sums=True))
        self.x, self.eps, self._t, self.sums = x, float(eps), None, sums

    def tensor(self) -> torch.Tensor:
        if self._t is None:
            self._t = row_stats(self.x, self.eps)
        return self._t


def conv3x3(x: torch.Tensor, w: torch.Tensor, *, x2: Optional[torch.Tensor] = None,
            bias: Optional[torch.Tensor] = None, rowbias: Optional[torch.Tensor] = None,
            rows_per_group: int = 0, residual: Optional[torch.Tensor] = None, stride: int = 1,
            upsample: bool = False, alpha: float = 1.0, post_scale: float = 1.0, act: int = ACT_NONE,
            out_f32: bool = False, pad_asym: bool = False, w_wino: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x: [images, H, W, Cin1] (+x2 [.., Cin2]); w: [Cout, 3, 3, Cin1+Cin2]; returns NHWC.
    pad_asym: pad (0 before, 1 after) instead of 1/1 -- diffusers Downsample2D(padding=0).
    w_wino: the weight once more in Winograd form [16, Cout, Cin] (layers.HipConv3x3._winograd_weight / ca_pack_w_wino): the library
    then runs the shapes it names as F(2x2, 3x3) (ca_conv_args.w_wino, ABI v12)."""
    _req_cuda(x, w, x2, bias, rowbias, residual, w_wino)
    assert x.dim() == 4 and x.is_contiguous() and w.dim() == 4 and w.is_contiguous()
    images, hin, win, cin1 = x.shape
    cin2 = 0
    if x2 is not None:
        assert x2.is_contiguous() and x2.shape[:3] == x.shape[:3] and x2.dtype == x.dtype
        cin2 = x2.shape[3]
    cout = w.shape[0]
    assert tuple(w.shape[1:]) == (3, 3, cin1 + cin2) and w.dtype == x.dtype
    hl, wl = (hin * 2, win * 2) if upsample else (hin, win)
    pad = 1 if pad_asym else 2
    hout, wout = (hl + pad - 3) // stride + 1, (wl + pad - 3) // stride + 1
    y = torch.empty((images, hout, wout, cout), device=x.device, dtype=torch.float32 if out_f32 else x.dtype)
    ld_res = 0
    if residual is not None:
        assert residual.dtype == x.dtype and residual.is_contiguous() and residual.numel() == images * hout * wout * cout, \
            f"residual {tuple(residual.shape)} {residual.dtype} contiguous={residual.is_contiguous()} vs output {(images, hout, wout, cout)} {x.dtype}"
        ld_res = cout
    if rowbias is not None:
        assert rowbias.dtype == torch.float32 and rowbias.dim() == 2 and rowbias.shape[1] == cout and rows_per_group > 0
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == cout
    args = ConvArgs(x=_p(x), x2=_p(x2), w=_p(w), y=_p(y), bias=_p(bias), rowbias=_p(rowbias),
                    residual=_p(residual), ld_res=ld_res,
                    ld_rowbias=rowbias.stride(0) if rowbias is not None else 0, images=images, hin=hin,
                    win=win, cin1=cin1, cin2=cin2, cout=cout, stride=stride, upsample=int(upsample),
                    rows_per_group=rows_per_group, alpha=alpha, post_scale=post_scale, act=act,
                    out_f32=int(out_f32), dtype=dt_code(x.dtype), pad_asym=int(pad_asym))
    if w_wino is not None and dispatch.conv_winograd:
        assert w_wino.dtype == x.dtype and w_wino.is_contiguous() and tuple(w_wino.shape) == (16, cout, cin1 + cin2)
        args.w_wino = _p(w_wino)
    wbytes = int(lib().ca_conv3x3_workspace_bytes(C.byref(args)))
    if wbytes > 0:  # split-K slabs for the small-M levels (allocator-cached, stream-ordered)
        ws = torch.empty((wbytes,), device=x.device, dtype=torch.uint8)
        args.workspace, args.workspace_bytes = _p(ws), wbytes
    _record_plan(lib().ca_conv3x3_plan_name, args)
    check(lib().ca_conv3x3(C.byref(args), _stream()), "ca_conv3x3")
    return y


_gn_wino_declined: dict = {}  # shapes group_norm_conv3x3_wino has declined (the library's answer depends on nothing else)


def group_norm_conv3x3_wino(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, w: torch.Tensor, w_wino: torch.Tensor, *,
                            x2: Optional[torch.Tensor] = None, groups: int = 32, eps: float = 1e-5, act: int = ACT_NONE,
                            bias: Optional[torch.Tensor] = None, rowbias: Optional[torch.Tensor] = None, rows_per_group: int = 0,
                            residual: Optional[torch.Tensor] = None, post_scale: float = 1.0) -> Optional[torch.Tensor]:
    """conv3x3(GroupNorm(+act)(cat(x, x2))) where the convolution takes the Winograd route AND the one-launch GroupNorm applies: the
    GroupNorm writes the transformed input V itself (ca_groupnorm_args.wino_v, ABI v12) and the convolution starts at its GEMM
    (ca_conv_args.x_is_wino_v) -- the normalised tensor is never written.  Per-image statistics only.  None where either side
    declines: the caller runs group_norm, then conv3x3."""
    if not (dispatch.conv_winograd and dispatch.gn_winograd):
        return None
    _req_cuda(x, gamma, beta, w, w_wino, x2, bias, rowbias, residual)
    if x.dim() != 4 or not x.is_contiguous() or x.dtype != torch.float16 or (x2 is not None and (not x2.is_contiguous() or x2.shape[:3] != x.shape[:3])):
        return None
    images, h, w_, c1 = x.shape
    c2 = 0 if x2 is None else x2.shape[3]
    c, cout = c1 + c2, w.shape[0]
    tiles = images * (h // 2) * (w_ // 2)
    # the cheap declines first (the probes only need non-null pointers: x stands in for V and y until both sides have accepted;
    # the decision is remembered per shape, so a resnet whose level the route declines costs two dictionary look-ups per eager step)
    key = (images, h, w_, c1, c2, cout, groups, act, rowbias is not None, residual is not None, rows_per_group)
    if _gn_wino_declined.get(key):
        return None
    cargs = ConvArgs(x=_p(x), x2=None, w=_p(w), y=_p(x), bias=_p(bias), rowbias=_p(rowbias), residual=_p(residual),
                     ld_res=cout if residual is not None else 0, ld_rowbias=rowbias.stride(0) if rowbias is not None else 0, images=images, hin=h,
                     win=w_, cin1=c, cin2=0, cout=cout, stride=1, upsample=0, rows_per_group=rows_per_group, alpha=1.0, post_scale=post_scale,
                     act=ACT_NONE, out_f32=0, dtype=dt_code(x.dtype), pad_asym=0, w_wino=_p(w_wino), x_is_wino_v=1)
    # would the convolution take the Winograd route (given enough workspace)?  ca_conv3x3_workspace_bytes alone cannot tell: it also
    # answers > 0 for the split-K plan of a shape the route declines
    cargs.workspace, cargs.workspace_bytes = _p(x), 1 << 60
    buf = C.create_string_buffer(64)
    if lib().ca_conv3x3_plan_name(C.byref(cargs), buf, 64) != 0 or not buf.value.decode().startswith("wino"):
        _gn_wino_declined[key] = True
        return None
    wbytes = int(lib().ca_conv3x3_workspace_bytes(C.byref(cargs)))
    gargs = GroupNormArgs(x=_p(x), x2=_p(x2), y=None, gamma=_p(gamma), beta=_p(beta), partials=None, images=images, hw=h * w_, c1=c1, c2=c2,
                          groups=groups, frames_per_stat=1, eps=eps, act=act, dtype=dt_code(x.dtype), wino_v=_p(x), wino_h=h, wino_w=w_)
    if wbytes <= 0 or not lib().ca_groupnorm_wino_supported(C.byref(gargs)):
        _gn_wino_declined[key] = True
        return None
    y = torch.empty((images, h, w_, cout), device=x.device, dtype=x.dtype)
    v = torch.empty((16, tiles, c), device=x.device, dtype=x.dtype)
    if residual is not None:
        assert residual.dtype == x.dtype and residual.is_contiguous() and residual.numel() == y.numel()
    ws = torch.empty((wbytes,), device=x.device, dtype=torch.uint8)
    cargs.x, cargs.y, gargs.wino_v = _p(v), _p(y), _p(v)
    cargs.workspace, cargs.workspace_bytes = _p(ws), wbytes
    if _plan_sink is not None:
        buf = C.create_string_buffer(64)
        _plan_sink.append("gn_" + (buf.value.decode() if lib().ca_conv3x3_plan_name(C.byref(cargs), buf, 64) == 0 else "?"))
    check(lib().ca_groupnorm(C.byref(gargs), _stream()), "ca_groupnorm(wino)")
    check(lib().ca_conv3x3(C.byref(cargs), _stream()), "ca_conv3x3(wino, V given)")
    return y


def group_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, *, x2: Optional[torch.Tensor] = None,
               groups: int = 32, frames_per_stat: int = 1, eps: float = 1e-5, act: int = ACT_NONE) -> torch.Tensor:
    """GroupNorm (+SiLU) over NHWC x (optionally channel-concatenated with x2)."""
    _req_cuda(x, gamma, beta, x2)
    assert x.dim() == 4 and x.is_contiguous()
    images, h, w_, c1 = x.shape
    c2 = 0
    if x2 is not None:
        assert x2.is_contiguous() and x2.shape[:3] == x.shape[:3] and x2.dtype == x.dtype
        c2 = x2.shape[3]
    c = c1 + c2
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.numel() == c and beta.numel() == c
    y = torch.empty((images, h, w_, c), device=x.device, dtype=x.dtype)
    nfl = lib().ca_groupnorm_partials_floats(images, h * w_, frames_per_stat, groups)
    partials = torch.empty((max(int(nfl), 1),), device=x.device, dtype=torch.float32)
    args = GroupNormArgs(x=_p(x), x2=_p(x2), y=_p(y), gamma=_p(gamma), beta=_p(beta), partials=_p(partials),
                         images=images, hw=h * w_, c1=c1, c2=c2, groups=groups,
                         frames_per_stat=frames_per_stat, eps=eps, act=act, dtype=dt_code(x.dtype))
    check(lib().ca_groupnorm(C.byref(args), _stream()), "ca_groupnorm")
    return y


def layer_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, *, pos: Optional[torch.Tensor] = None,
               rows_per_frame: int = 1, frames: int = 1, eps: float = 1e-5) -> torch.Tensor:
    _req_cuda(x, gamma, beta, pos)
    assert x.dim() == 2 and x.is_contiguous()
    rows, c = x.shape
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32
    if pos is not None:
        assert pos.dtype == torch.float32 and pos.is_contiguous() and pos.shape == (frames, c)
    y = torch.empty_like(x)
    args = LayerNormArgs(x=_p(x), y=_p(y), gamma=_p(gamma), beta=_p(beta), pos=_p(pos), rows=rows, c=c,
                         rows_per_frame=rows_per_frame, frames=frames, eps=eps, dtype=dt_code(x.dtype))
    check(lib().ca_layernorm(C.byref(args), _stream()), "ca_layernorm")
    return y


def row_stats(x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """(mean, rstd) of every row of x [rows, C] as fp32 [rows, 2]: the statistics of a LayerNorm that is
    folded into the following GEMM (gemm(..., ln=(stats, colsum)))."""
    _req_cuda(x)
    assert x.dim() == 2 and x.is_contiguous()
    rows, c = x.shape
    st = torch.empty((rows, 2), device=x.device, dtype=torch.float32)
    args = LayerNormArgs(x=_p(x), y=None, gamma=None, beta=None, pos=None, rows=rows, c=c, rows_per_frame=1, frames=1,
                         eps=eps, dtype=dt_code(x.dtype), stats=_p(st))
    check(lib().ca_layernorm(C.byref(args), _stream()), "ca_layernorm(stats)")
    return st


def attention_raw(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, o: torch.Tensor, *, q_off: int, k_off: int,
                  v_off: int, o_off: int, q_strides, o_strides, k_strides, inner_count: int, kv_inner_count: int,
                  kv_div: int, batches: int, heads: int, head_dim: int, nq: int, nk: int, scale: float,
                  out_scale: float = 1.0, accumulate: bool = False, kv_mod: int = 0, causal: bool = False,
                  key_mask: Optional[torch.Tensor] = None) -> None:
    """Direct mapping of ca_attention. *_off are element offsets into the given storage tensors;
    *_strides = (outer, inner, row) in elements.  key_mask: uint8 [batches, nk], 0 = key invisible (ca_attn_args.key_mask)."""
    _req_cuda(q, k, v, o, key_mask)
    if key_mask is not None:
        assert key_mask.dtype == torch.uint8 and key_mask.dim() == 2 and key_mask.shape == (batches, nk) and key_mask.stride(1) == 1
    es = q.element_size()
    args = AttnArgs(q=q.data_ptr() + q_off * es, k=k.data_ptr() + k_off * es, v=v.data_ptr() + v_off * es,
                    o=o.data_ptr() + o_off * es,
                    q_outer=q_strides[0], q_inner=q_strides[1], q_row=q_strides[2],
                    o_outer=o_strides[0], o_inner=o_strides[1], o_row=o_strides[2],
                    k_outer=k_strides[0], k_inner=k_strides[1], k_row=k_strides[2],
                    inner_count=inner_count, kv_inner_count=kv_inner_count, kv_div=kv_div,
                    kv_mod=kv_mod if kv_mod > 0 else batches, batches=batches, heads=heads, head_dim=head_dim, nq=nq, nk=nk, scale=scale, out_scale=out_scale,
                    accumulate=int(accumulate), dtype=dt_code(q.dtype), causal=int(causal),
                    key_mask=_p(key_mask), key_mask_stride=key_mask.stride(0) if key_mask is not None else 0)
    _record_plan(lib().ca_attention_plan_name, args)
    check(lib().ca_attention(C.byref(args), _stream()), "ca_attention")


def attention_spatial(qkv: torch.Tensor, images: int, tokens: int, heads: int, causal: bool = False,
                      key_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Self-attention per image. qkv: [images*tokens, 3C] (q | k | v); returns [images*tokens, C].
    key_mask: uint8 [images, tokens], 0 = that token is invisible as a key (transformers' attention_mask)."""
    c = qkv.shape[1] // 3
    d = c // heads
    o = torch.empty((images * tokens, c), device=qkv.device, dtype=qkv.dtype)
    ld = qkv.stride(0)
    attention_raw(qkv, qkv, qkv, o, q_off=0, k_off=c, v_off=2 * c, o_off=0,
                  q_strides=(tokens * ld, 0, ld), o_strides=(tokens * c, 0, c), k_strides=(tokens * ld, 0, ld),
                  inner_count=1, kv_inner_count=1, kv_div=1, batches=images, heads=heads, head_dim=d,
                  nq=tokens, nk=tokens, scale=d ** -0.5, causal=causal, key_mask=key_mask)
    return o


def attention_cross(q: torch.Tensor, kv: torch.Tensor, images: int, tokens: int, heads: int, kv_tokens: int,
                    kv_rows_per_batch: int, frames_per_kv: int, *, out: Optional[torch.Tensor] = None,
                    out_scale: float = 1.0, accumulate: bool = False, kv_row_offset: int = 0,
                    kv_mod: int = 0) -> torch.Tensor:
    """Cross-attention. q: [images*tokens, C]; kv: [kv_batches*kv_rows_per_batch, 2C] (k | v).
    Image z uses kv batch (z // frames_per_kv) % kv_mod, rows [kv_row_offset, kv_row_offset + kv_tokens)."""
    c = q.shape[1]
    d = c // heads
    if out is None:
        out = torch.empty_like(q)
    ldk = kv.stride(0)
    attention_raw(q, kv, kv, out, q_off=0, k_off=kv_row_offset * ldk, v_off=kv_row_offset * ldk + c, o_off=0,
                  q_strides=(tokens * q.stride(0), 0, q.stride(0)), o_strides=(tokens * out.stride(0), 0, out.stride(0)),
                  k_strides=(kv_rows_per_batch * ldk, 0, ldk), inner_count=1, kv_inner_count=1,
                  kv_div=frames_per_kv, batches=images, heads=heads, head_dim=d, nq=tokens, nk=kv_tokens,
                  scale=d ** -0.5, out_scale=out_scale, accumulate=accumulate, kv_mod=kv_mod)
    return out


def attention_temporal(qkv: torch.Tensor, b: int, frames: int, tokens: int, heads: int) -> torch.Tensor:
    """Attention over the frame axis. qkv rows are in (b f n) order: [b*frames*tokens, 3C]."""
    c = qkv.shape[1] // 3
    d = c // heads
    o = torch.empty((b * frames * tokens, c), device=qkv.device, dtype=qkv.dtype)
    ld = qkv.stride(0)
    attention_raw(qkv, qkv, qkv, o, q_off=0, k_off=c, v_off=2 * c, o_off=0,
                  q_strides=(frames * tokens * ld, ld, tokens * ld), o_strides=(frames * tokens * c, c, tokens * c),
                  k_strides=(frames * tokens * ld, ld, tokens * ld), inner_count=tokens, kv_inner_count=tokens,
                  kv_div=1, batches=b * tokens, heads=heads, head_dim=d, nq=frames, nk=frames, scale=d ** -0.5)
    return o


def add_bcast(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """a + b where b is tiled along the leading dimension (b.numel() divides a.numel())."""
    _req_cuda(a, b)
    assert a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype and a.numel() % b.numel() == 0
    if out is None:
        out = torch.empty_like(a)
    check(lib().ca_add_bcast(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), b.numel(), dt_code(a.dtype),
                             _stream()), "ca_add_bcast")
    return out


def repeat_batch(x: torch.Tensor, times: int = 2) -> torch.Tensor:
    """torch.cat([x] * times) along dim 0 with one read of x (ca_repeat, ABI v8; context.dispatch.repeat_kernel = False: torch.cat, for A/B runs)."""
    if not dispatch.repeat_kernel:
        return torch.cat([x] * times)
    _req_cuda(x)
    if not x.is_contiguous():
        x = x.contiguous()  # (a strided view: one gather first; sizes / alignments the 16-byte kernel cannot take go byte-wise in the library)
    if x.numel() == 0:
        return x.new_empty((times * x.shape[0],) + tuple(x.shape[1:]))
    out = torch.empty((times * x.shape[0],) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
    check(lib().ca_repeat(x.data_ptr(), out.data_ptr(), x.numel() * x.element_size(), times, _stream()), "ca_repeat")
    return out


def softmax_rows(x: torch.Tensor, dtype: torch.dtype, scale: float = 1.0) -> torch.Tensor:
    """Row softmax of an fp32 [rows, cols] score matrix (last dim contiguous), output in `dtype`."""
    _req_cuda(x)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    y = torch.empty(x.shape, device=x.device, dtype=dtype)
    check(lib().ca_softmax_rows(x.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], x.stride(0), y.stride(0), float(scale),
                                dt_code(dtype), _stream()), "ca_softmax_rows")
    return y


def silu_f32(x: torch.Tensor) -> torch.Tensor:
    _req_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    y = torch.empty_like(x)
    check(lib().ca_silu_f32(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "ca_silu_f32")
    return y


def timestep_embedding(t, batch: int, dim: int, dtype: torch.dtype, device) -> torch.Tensor:
    """t: python number (same for the whole batch) or fp32 device tensor [batch]."""
    out = torch.empty((batch, dim), device=device, dtype=dtype)
    if isinstance(t, torch.Tensor):
        _req_cuda(t)
        assert t.dtype == torch.float32 and t.numel() == batch
        check(lib().ca_timestep_embedding(t.data_ptr(), 0.0, out.data_ptr(), batch, dim, dt_code(dtype), _stream()),
              "ca_timestep_embedding")
    else:
        check(lib().ca_timestep_embedding(None, float(t), out.data_ptr(), batch, dim, dt_code(dtype), _stream()),
              "ca_timestep_embedding")
    return out


def latents_to_nhwc(latents: torch.Tensor, cpad: int, rep: int, in_scale: float, dtype: torch.dtype,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _req_cuda(latents)
    assert latents.dtype == torch.float32 and latents.is_contiguous() and latents.dim() == 5
    b0, c, f, h, w = latents.shape
    if out is None:
        out = torch.empty((rep * b0 * f, h, w, cpad), device=latents.device, dtype=dtype)
    assert out.shape == (rep * b0 * f, h, w, cpad) and out.dtype == dtype and out.is_contiguous()
    check(lib().ca_latents_to_nhwc(latents.data_ptr(), out.data_ptr(), b0, c, f, h, w, cpad, rep, in_scale,
                                   dt_code(dtype), _stream()), "ca_latents_to_nhwc")
    return out


def nhwc_to_ncfhw_f32(x: torch.Tensor, b: int, c: int, f: int) -> torch.Tensor:
    """x: [b*f, h, w, ldx] (act dtype or fp32) -> [b, c, f, h, w] fp32."""
    _req_cuda(x)
    assert x.dim() == 4 and x.is_contiguous()
    _, h, w, ldx = x.shape
    out = torch.empty((b, c, f, h, w), device=x.device, dtype=torch.float32)
    is_f32 = x.dtype == torch.float32
    check(lib().ca_nhwc_to_ncfhw_f32(x.data_ptr(), out.data_ptr(), b, c, f, h, w, ldx, int(is_f32),
                                     CA_BF16 if is_f32 else dt_code(x.dtype), _stream()), "ca_nhwc_to_ncfhw_f32")
    return out


_KIND = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def ncfhw_to_nhwc(x: torch.Tensor, cpad: int, dtype: torch.dtype) -> torch.Tensor:
    """x: [b, c, f, h, w] any strides, fp32/fp16/bf16 -> [b*f, h, w, cpad] `dtype` (zero padded)."""
    _req_cuda(x)
    assert x.dim() == 5 and x.dtype in _KIND
    b, c, f, h, w = x.shape
    out = torch.empty((b * f, h, w, cpad), device=x.device, dtype=dtype)
    st = (C.c_int64 * 5)(*x.stride())
    check(lib().ca_ncfhw_to_nhwc(x.data_ptr(), _KIND[x.dtype], st, out.data_ptr(), b, c, f, h, w, cpad,
                                 dt_code(dtype), _stream()), "ca_ncfhw_to_nhwc")
    return out


def cfg_scheduler_step(eps: torch.Tensor, rep: int, guidance: float, latents: torch.Tensor,
                       noise: Optional[torch.Tensor], coef, clip: float = 0.0, want_denoised: bool = False):
    """eps: NHWC fp32 [rep*f, h, w, ld]; latents/noise: [1, c, f, h, w] fp32. Returns (prev, denoised|None)."""
    _req_cuda(eps, latents, noise)
    assert eps.dtype == torch.float32 and eps.is_contiguous() and latents.dtype == torch.float32 and latents.is_contiguous()
    _, c, f, h, w = latents.shape
    assert latents.shape[0] == 1 and eps.shape[0] == rep * f
    if noise is not None:
        assert noise.dtype == torch.float32 and noise.is_contiguous() and noise.shape == latents.shape
    prev = torch.empty_like(latents)
    den = torch.empty_like(latents) if want_denoised else None
    cf = (C.c_float * 7)(*[float(v) for v in coef])
    check(lib().ca_cfg_scheduler_step(eps.data_ptr(), eps.shape[3], rep, float(guidance), latents.data_ptr(),
                                      _p(noise), prev.data_ptr(), _p(den), c, f, h, w, cf, float(clip), _stream()),
          "ca_cfg_scheduler_step")
    return prev, den


def lincomb(terms, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = sum_k coef_k * x_k for fp32 tensors of one shape; terms: [(tensor, coef), ...] (1..8). ca_lincomb."""
    xs = [t for t, _ in terms]
    _req_cuda(*xs)
    assert 1 <= len(terms) <= 8 and all(t.dtype == torch.float32 and t.is_contiguous() and t.shape == xs[0].shape for t in xs)
    if out is None:
        out = torch.empty_like(xs[0])
    ptrs = (C.c_void_p * len(xs))(*[t.data_ptr() for t in xs])
    cf = (C.c_float * len(xs))(*[float(c) for _, c in terms])
    check(lib().ca_lincomb(out.data_ptr(), ptrs, cf, len(xs), xs[0].numel(), _stream()), "ca_lincomb")
    return out


def cfg_combined_eps(eps: torch.Tensor, rep: int, guidance: float, latents: torch.Tensor) -> torch.Tensor:
    """The classifier-free-guidance combine alone: eps NHWC fp32 [rep*f,h,w,ld] -> [1,c,f,h,w] fp32
    (ca_cfg_scheduler_step with prev := eps; the multistep samplers keep it as history)."""
    e, _ = cfg_scheduler_step(eps, rep, guidance, latents, None, [0.0, 1.0, 0.0, 0.0, 0.0, 1.0, 0.0], 0.0)
    return e
