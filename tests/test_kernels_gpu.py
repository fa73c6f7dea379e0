"""Kernel-level parity: every C-ABI entry point vs a plain torch fp32 reference of the same op.

Inputs are rounded to the activation dtype first, so the only differences are fp32-accumulation
order and the final rounding of the output (bf16: 2^-9 relative).
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
DTYPES = [torch.bfloat16, torch.float16]


def _k():
    from controlanimate_amd import kernels
    return kernels


def rnd(*shape, dtype=torch.bfloat16, scale=1.0, seed=None):
    g = torch.Generator().manual_seed(seed if seed is not None else (hash(shape) % 100003))
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def close(out, ref, dtype, what, rel=None):
    out = out.float().cpu()
    ref = ref.float()
    assert out.shape == ref.shape, f"{what}: shape {tuple(out.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(out).all(), f"{what}: non-finite output"
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -10
    rel = rel if rel is not None else 2.5 * eps
    err = (out - ref).abs()
    denom = ref.abs().max().clamp_min(1e-6)
    rel_l2 = (out - ref).norm() / ref.norm().clamp_min(1e-12)
    max_rel = (err.max() / denom).item()
    assert rel_l2.item() < rel and max_rel < 4 * rel, f"{what}: rel_l2={rel_l2.item():.3e} max_err/max_ref={max_rel:.3e} (tol {rel:.1e})"


# ------------------------------------------------------------------------------------ GEMM
GEMM_CASES = [
    # m, n, k1, k2, bias, rowbias, residual, alpha, post, act, geglu, out_f32
    (256, 128, 64, 0, False, False, False, 1.0, 1.0, 0, False, False),
    (200, 320, 320, 0, True, False, True, 1.0, 1.0, 0, False, False),
    (130, 24, 40, 0, True, False, False, 1.0, 1.0, 0, False, False),
    (512, 640, 320, 0, True, True, True, 0.5, 0.7, 1, False, False),
    (300, 256, 64, 0, True, False, False, 1.0, 1.0, 0, True, False),
    (300, 2560, 320, 0, True, False, False, 1.0, 1.0, 0, True, False),
    (256, 320, 320, 640, True, False, False, 1.0, 1.0, 0, False, False),
    (2, 1280, 320, 0, True, False, False, 1.0, 1.0, 1, False, True),
    (154, 640, 768, 0, False, False, False, 1.0, 1.0, 0, False, False),
    (4096, 1280, 1280, 0, True, False, True, 1.0, 1.0, 0, False, False),
    (1000, 8, 64, 0, True, False, False, 1.0, 1.0, 0, False, False),
    (77, 4, 32, 0, False, False, False, 1.0, 1.0, 0, False, True),
    # grids of >= 512 blocks take the single-LDS-buffer / 4-blocks-per-CU pipeline
    (16384, 320, 320, 0, True, False, True, 1.0, 1.0, 0, False, False),
    (16384, 2560, 320, 0, True, False, False, 1.0, 1.0, 0, True, False),
    (16384, 960, 320, 0, False, False, False, 1.0, 1.0, 0, False, False),
    (16384, 1280, 1280, 0, True, True, True, 1.0, 1.0, 0, False, False),
    (65536, 320, 640, 640, True, False, False, 1.0, 1.0, 0, False, False),
    # <= 192 tiles of 128x128 with >= 16 K tiles: the split-K slab schedule of the 8x8-latent level (ABI v6 workspace)
    (2048, 1280, 1280, 0, True, False, True, 1.0, 1.0, 0, False, False),
    (2040, 1280, 5120, 0, True, True, True, 0.5, 0.7, 1, False, False),
    (300, 1280, 1024, 1024, True, False, False, 1.0, 1.0, 0, False, False),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", GEMM_CASES)
def test_gemm(case, dtype):
    k = _k()
    m, n, k1, k2, has_bias, has_rb, has_res, alpha, post, act, geglu, out_f32 = case
    a = rnd(m, k1, dtype=dtype, seed=1)
    a2 = rnd(m, k2, dtype=dtype, seed=2) if k2 else None
    w = rnd(n, k1 + k2, dtype=dtype, scale=(k1 + k2) ** -0.5, seed=3)
    bias = rnd(n, dtype=torch.float32, seed=4) if has_bias else None
    rpg = 64
    rb = rnd((m + rpg - 1) // rpg, n, dtype=torch.float32, seed=5) if has_rb else None
    res = rnd(m, n, dtype=dtype, seed=6) if has_res else None
    acat = a.float() if a2 is None else torch.cat([a.float(), a2.float()], 1)
    ref = acat @ w.float().t()
    if bias is not None:
        ref = ref + bias
    if rb is not None:
        ref = ref + rb[torch.arange(m) // rpg]
    ref = ref * alpha
    if res is not None:
        ref = ref + res.float()
    ref = ref * post
    if act == 1:
        ref = F.silu(ref)
    if geglu:
        ref = ref[:, 0::2] * F.gelu(ref[:, 1::2])
    out = k.gemm(a.to(DEV), w.to(DEV), a2=None if a2 is None else a2.to(DEV),
                 bias=None if bias is None else bias.to(DEV), rowbias=None if rb is None else rb.to(DEV),
                 rows_per_group=rpg if has_rb else 0, residual=None if res is None else res.to(DEV),
                 alpha=alpha, post_scale=post, act=act, geglu=geglu, out_f32=out_f32)
    torch.cuda.synchronize()
    assert out.dtype == (torch.float32 if out_f32 else dtype)
    close(out, ref, dtype, f"gemm{case}")


def test_gemm_strided_views():
    """lda/ldc larger than the logical width (column slices of a wider buffer)."""
    k = _k()
    dtype = torch.bfloat16
    big = rnd(300, 960, dtype=dtype, seed=7).to(DEV)
    w = rnd(320, 320, dtype=dtype, scale=320 ** -0.5, seed=8).to(DEV)
    a = big[:, 320:640]
    outbuf = torch.zeros(300, 640, device=DEV, dtype=dtype)
    out = outbuf[:, 320:]
    k.gemm(a, w, out=out)
    torch.cuda.synchronize()
    close(out, a.float().cpu() @ w.float().cpu().t(), dtype, "gemm strided")
    assert outbuf[:, :320].abs().max().item() == 0


def test_gemm_identity_asymmetric():
    """A = I with an asymmetric W catches a transposed C write."""
    k = _k()
    dtype = torch.bfloat16
    n, kk = 192, 128
    a = torch.eye(kk, dtype=dtype)
    w = (torch.arange(n)[:, None] * 0.01 + torch.arange(kk)[None, :] * 0.0001).to(dtype)
    out = k.gemm(a.to(DEV), w.to(DEV))
    torch.cuda.synchronize()
    close(out, w.float().t(), dtype, "gemm identity")


def test_gemm_rejects_bad_args():
    k = _k()
    from controlanimate_amd._capi import CAHipError
    a = rnd(16, 12, seed=1).to(DEV)  # K not a multiple of 8
    w = rnd(16, 12, seed=2).to(DEV)
    with pytest.raises(CAHipError):
        k.gemm(a, w)


# ------------------------------------------------------------------------------------ conv
CONV_CASES = [
    # images, h, w, cin1, cin2, cout, stride, upsample, bias, rowbias, residual, act, out_f32
    (2, 16, 16, 32, 0, 64, 1, False, True, False, False, 0, False),
    (2, 16, 16, 64, 0, 64, 2, False, True, False, False, 0, False),
    (3, 10, 6, 32, 0, 32, 1, False, True, True, True, 0, False),
    (3, 9, 7, 32, 0, 32, 2, False, True, False, False, 0, False),
    (2, 8, 8, 64, 0, 64, 1, True, True, False, False, 0, False),
    (2, 8, 8, 64, 32, 128, 1, False, True, True, True, 0, False),
    (4, 16, 16, 8, 0, 320, 1, False, True, False, False, 0, False),
    (4, 16, 16, 320, 0, 4, 1, False, True, False, False, 0, True),
    (2, 32, 32, 8, 0, 16, 1, False, True, False, False, 1, False),
    (2, 16, 16, 96, 0, 256, 2, False, True, False, False, 1, False),
    (2, 8, 8, 1280, 1280, 1280, 1, False, True, True, False, 0, False),
    (8, 32, 32, 320, 0, 320, 1, False, True, True, True, 0, False),
    # >= 512 blocks: single-LDS-buffer pipeline
    (16, 64, 64, 320, 0, 320, 1, False, True, True, True, 0, False),
    (16, 32, 32, 640, 320, 640, 1, False, True, False, False, 0, False),
    (16, 32, 32, 640, 0, 640, 1, True, True, False, False, 0, False),
    (16, 64, 64, 320, 0, 320, 2, False, True, False, False, 0, False),
    # small M, long K: split-K slabs + reduce kernel (epilogue variants)
    (32, 8, 8, 1280, 0, 1280, 1, False, True, True, True, 1, False),
    (4, 8, 8, 640, 0, 128, 1, False, True, False, False, 0, True),
    (8, 16, 16, 640, 640, 256, 1, False, False, False, True, 0, False),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3x3(case, dtype):
    k = _k()
    images, h, w_, c1, c2, cout, stride, ups, has_bias, has_rb, has_res, act, out_f32 = case
    cin = c1 + c2
    x = rnd(images, h, w_, c1, dtype=dtype, seed=11)
    x2 = rnd(images, h, w_, c2, dtype=dtype, seed=12) if c2 else None
    wt = rnd(cout, cin, 3, 3, dtype=dtype, scale=(9 * cin) ** -0.5, seed=13)  # torch OIHW
    bias = rnd(cout, dtype=torch.float32, seed=14) if has_bias else None
    xin = x.float() if x2 is None else torch.cat([x.float(), x2.float()], -1)
    xin = xin.permute(0, 3, 1, 2)
    if ups:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xin, wt.float(), bias, stride=stride, padding=1)
    hout, wout = ref.shape[2:]
    frames = 1 if images % 2 else 2  # rowbias per batch element: images = b * frames
    rpg = frames * hout * wout
    rb = rnd(images // frames, cout, dtype=torch.float32, seed=15) if has_rb else None
    if rb is not None:
        ref = ref + rb.repeat_interleave(frames, 0)[:, :, None, None]
    res = rnd(images, hout, wout, cout, dtype=dtype, seed=16) if has_res else None
    if res is not None:
        ref = (ref + res.float().permute(0, 3, 1, 2)) * 0.5
    if act == 1:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 3, 1)
    out = k.conv3x3(x.to(DEV), wt.permute(0, 2, 3, 1).contiguous().to(DEV), x2=None if x2 is None else x2.to(DEV),
                    bias=None if bias is None else bias.to(DEV), rowbias=None if rb is None else rb.to(DEV),
                    rows_per_group=rpg if has_rb else 0, residual=None if res is None else res.to(DEV),
                    stride=stride, upsample=ups, post_scale=0.5 if has_res else 1.0, act=act, out_f32=out_f32)
    torch.cuda.synchronize()
    close(out, ref, dtype, f"conv{case}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 16, 16, 64, 128), (3, 10, 6, 32, 32), (2, 64, 64, 128, 128)])
def test_conv3x3_asymmetric_pad_stride2(case, dtype):
    """diffusers Downsample2D(padding=0): F.pad(x, (0,1,0,1)) + Conv2d(3, stride=2) (VAE encoder)."""
    k = _k()
    images, h, w_, cin, cout = case
    x = rnd(images, h, w_, cin, dtype=dtype, seed=21)
    wt = rnd(cout, cin, 3, 3, dtype=dtype, scale=(9 * cin) ** -0.5, seed=22)
    bias = rnd(cout, dtype=torch.float32, seed=23)
    ref = F.conv2d(F.pad(x.float().permute(0, 3, 1, 2), (0, 1, 0, 1)), wt.float(), bias, stride=2, padding=0).permute(0, 2, 3, 1)
    out = k.conv3x3(x.to(DEV), wt.permute(0, 2, 3, 1).contiguous().to(DEV), bias=bias.to(DEV), stride=2, pad_asym=True)
    torch.cuda.synchronize()
    close(out, ref, dtype, f"conv_asym{case}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,cols", [(7, 64), (300, 4096), (16, 1000)])
def test_softmax_rows(rows, cols, dtype):
    k = _k()
    x = rnd(rows, cols, dtype=torch.float32, scale=4.0, seed=31)
    ref = torch.softmax(x * 0.37, -1)
    out = k.softmax_rows(x.to(DEV), dtype, 0.37)
    torch.cuda.synchronize()
    close(out, ref, dtype, f"softmax_rows({rows},{cols})")
    assert (out.float().sum(-1).cpu() - 1).abs().max() < (2e-2 if dtype == torch.bfloat16 else 4e-3)


# ------------------------------------------------------------------------------------ norms
GN_CASES = [
    # images, h, w, c1, c2, frames_per_stat, act, eps
    (4, 8, 8, 320, 0, 1, 1, 1e-5),
    (4, 8, 8, 320, 0, 2, 1, 1e-5),
    (2, 16, 16, 32, 0, 1, 0, 1e-6),
    (2, 5, 7, 64, 0, 1, 0, 1e-6),
    (2, 8, 8, 1280, 1280, 1, 1, 1e-5),
    (2, 8, 8, 640, 320, 2, 1, 1e-5),
    (16, 64, 64, 320, 0, 1, 1, 1e-5),
    (8, 32, 32, 320, 0, 8, 1, 1e-5),
    # one-launch kernels (ca_groupnorm): a block owns 40 channels = 4 / 2 / 1 norm groups of one image
    (3, 64, 64, 320, 0, 1, 1, 1e-5),
    (8, 32, 32, 640, 0, 1, 1, 1e-6),
    (8, 32, 32, 320, 320, 1, 1, 1e-5),
    (16, 16, 16, 1280, 1280, 1, 0, 1e-5),
    (3, 24, 24, 1280, 0, 1, 1, 1e-5),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", GN_CASES)
def test_groupnorm(case, dtype):
    k = _k()
    images, h, w_, c1, c2, fps, act, eps = case
    c = c1 + c2
    x = (rnd(images, h, w_, c1, dtype=torch.float32, seed=21) * 1.5 + 0.3).to(dtype)
    x2 = (rnd(images, h, w_, c2, dtype=torch.float32, seed=22) * 0.7 - 0.2).to(dtype) if c2 else None
    gamma = rnd(c, dtype=torch.float32, seed=23) * 0.2 + 1.0
    beta = rnd(c, dtype=torch.float32, seed=24) * 0.2
    xin = x.float() if x2 is None else torch.cat([x.float(), x2.float()], -1)
    # [images, h, w, c] -> [images/fps, c, fps, h, w]: stats span fps frames
    x5 = xin.reshape(images // fps, fps, h, w_, c).permute(0, 4, 1, 2, 3)
    ref = F.group_norm(x5, 32, gamma, beta, eps)
    if act:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 3, 4, 1).reshape(images, h, w_, c)
    out = k.group_norm(x.to(DEV), gamma.to(DEV), beta.to(DEV), x2=None if x2 is None else x2.to(DEV),
                       frames_per_stat=fps, eps=eps, act=act)
    torch.cuda.synchronize()
    close(out, ref, dtype, f"groupnorm{case}")


def test_groupnorm_deterministic():
    k = _k()
    x = rnd(8, 32, 32, 320, seed=25).to(DEV)
    g = torch.ones(320, device=DEV)
    b = torch.zeros(320, device=DEV)
    o1 = k.group_norm(x, g, b, act=1)
    o2 = k.group_norm(x, g, b, act=1)
    torch.cuda.synchronize()
    assert torch.equal(o1, o2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,c,frames,rpf", [(100, 320, 0, 0), (64, 1280, 0, 0), (33, 32, 0, 0), (2 * 16 * 12, 640, 16, 12),
                                               (5000, 320, 8, 125)])
def test_layernorm(rows, c, frames, rpf, dtype):
    k = _k()
    x = (rnd(rows, c, dtype=torch.float32, seed=31) * 2 + 0.5).to(dtype)
    gamma = rnd(c, dtype=torch.float32, seed=32) * 0.2 + 1.0
    beta = rnd(c, dtype=torch.float32, seed=33) * 0.2
    ref = F.layer_norm(x.float(), (c,), gamma, beta, 1e-5)
    pos = None
    if frames:
        pos = rnd(frames, c, dtype=torch.float32, seed=34)
        fr = (torch.arange(rows) // rpf) % frames
        ref = ref + pos[fr]
    out = k.layer_norm(x.to(DEV), gamma.to(DEV), beta.to(DEV), pos=None if pos is None else pos.to(DEV),
                       rows_per_frame=rpf or 1, frames=frames or 1)
    torch.cuda.synchronize()
    close(out, ref, dtype, f"layernorm({rows},{c},{frames})")


# ------------------------------------------------------------------------------------ attention
def _sdpa(q, k_, v):  # [B, H, N, d] fp32
    s = (q @ k_.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
    return torch.softmax(s, -1) @ v


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("images,tokens,heads,d", [(2, 256, 8, 40), (2, 100, 8, 80), (1, 64, 8, 160), (3, 16, 4, 8),
                                                  (2, 4, 8, 16), (2, 1, 8, 32), (1, 1024, 8, 40), (1, 300, 2, 64),
                                                  (1, 200, 2, 128), (2, 144, 8, 160),
                                                  # (d = 80, >= 256 keys: the LDS-DMA kernel with 256-byte rows, ragged and whole tiles)
                                                  (1, 1024, 8, 80), (2, 300, 8, 80), (2, 256, 4, 80)])
def test_attention_spatial(images, tokens, heads, d, dtype):
    k = _k()
    c = heads * d
    qkv = rnd(images * tokens, 3 * c, dtype=dtype, seed=41)
    q, kk, v = [t.float().reshape(images, tokens, heads, d).transpose(1, 2) for t in qkv.split(c, dim=1)]
    ref = _sdpa(q, kk, v).transpose(1, 2).reshape(images * tokens, c)
    out = k.attention_spatial(qkv.to(DEV), images, tokens, heads)
    torch.cuda.synchronize()
    close(out, ref, dtype, f"attn_spatial({images},{tokens},{heads},{d})", rel=8e-3 if dtype == torch.bfloat16 else 3e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("images,tokens,heads,d", [(2, 77, 12, 64), (1, 16, 4, 16), (2, 130, 2, 80)])
def test_attention_causal(images, tokens, heads, d, dtype):
    """CLIP text encoder: key j is visible to query i only if j <= i."""
    k = _k()
    c = heads * d
    qkv = rnd(images * tokens, 3 * c, dtype=dtype, seed=43)
    q, kk, v = [t.float().reshape(images, tokens, heads, d).transpose(1, 2) for t in qkv.split(c, dim=1)]
    ref = F.scaled_dot_product_attention(q, kk, v, is_causal=True).transpose(1, 2).reshape(images * tokens, c)
    out = k.attention_spatial(qkv.to(DEV), images, tokens, heads, causal=True)
    torch.cuda.synchronize()
    close(out, ref, dtype, f"attn_causal({images},{tokens},{heads},{d})", rel=8e-3 if dtype == torch.bfloat16 else 3e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("m,n,k,geglu,rb", [(300, 960, 320, False, False), (4096, 2560, 320, True, False), (512, 640, 640, False, True),
                                             (77, 1280, 1280, False, False), (2048, 1280, 3072, False, False)])  # last: 48 K tiles on 160 tiles -> split-K reduce
def test_gemm_with_folded_layernorm(m, n, k, geglu, rb, dtype):
    """LN(x) W^T + b == rstd * (x W'^T - mean * colsum(W')) + (W beta + b), W' = W diag(gamma):
    ca_layernorm(stats) + ca_gemm(ln_stats, ln_colsum) against LayerNorm -> Linear in fp32."""
    k_ = _k()
    x = (rnd(m, k, dtype=torch.float32, seed=61) * 1.7 + 0.8).to(dtype)     # non-zero mean: the correction term matters
    w = rnd(n, k, dtype=torch.float32, scale=k ** -0.5, seed=62)
    b = rnd(n, dtype=torch.float32, seed=63)
    gamma, beta = 1 + 0.2 * rnd(k, dtype=torch.float32, seed=64), 0.3 * rnd(k, dtype=torch.float32, seed=65)
    rowbias = rnd(4, n, dtype=torch.float32, seed=66) if rb else None
    y = F.linear(F.layer_norm(x.float(), (k,), gamma, beta, 1e-5), w, b)
    if rb:
        y = y + rowbias.repeat_interleave(m // 4, 0)
    if geglu:
        h, g = y.chunk(2, -1)
        y = h * F.gelu(g)
    wf = (w * gamma[None, :])
    bf = w @ beta + b
    if geglu:
        from controlanimate_amd.layers import geglu_interleave
        wf, bf = geglu_interleave(wf), geglu_interleave(bf)
    wp = wf.to(dtype)
    cs = wp.float().sum(1)
    xd = x.to(DEV)
    ref_st = torch.stack([x.float().mean(1), (x.float().var(1, unbiased=False) + 1e-5).rsqrt()], 1)
    if k <= 2048:
        st = k_.row_stats(xd, 1e-5)
        assert torch.allclose(st.cpu(), ref_st, rtol=2e-4, atol=2e-5)
    else:  # (the statistics kernel stops at 2048 channels; the GEMM takes caller-computed statistics of any width)
        st = ref_st.contiguous().to(DEV)
    out = k_.gemm(xd, wp.to(DEV), bias=bf.to(DEV), geglu=geglu, ln=(st, cs.to(DEV)),
                  rowbias=None if rowbias is None else rowbias.to(DEV), rows_per_group=m // 4 if rb else 0)
    torch.cuda.synchronize()
    close(out, y, dtype, f"gemm_ln({m},{n},{k})")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("act", [2, 3])
def test_gemm_clip_activations(act, dtype):
    """CA_ACT_QUICK_GELU (x * sigmoid(1.702 x)) and CA_ACT_GELU (erf) epilogues."""
    k = _k()
    a = rnd(154, 768, dtype=dtype, seed=51)
    w = rnd(3072, 768, dtype=dtype, scale=768 ** -0.5, seed=52)
    b = rnd(3072, dtype=torch.float32, seed=53)
    y = a.float() @ w.float().t() + b
    ref = y * torch.sigmoid(1.702 * y) if act == 2 else F.gelu(y)
    out = k.gemm(a.to(DEV), w.to(DEV), bias=b.to(DEV), act=act)
    torch.cuda.synchronize()
    close(out, ref, dtype, f"gemm act {act}")


def test_attention_online_softmax_rescale():
    """A spiked key in a LATER kv block forces the running-max rescale branch."""
    k = _k()
    dtype = torch.bfloat16
    images, tokens, heads, d = 1, 256, 1, 64
    c = heads * d
    qkv = rnd(images * tokens, 3 * c, dtype=dtype, seed=42).float()
    qkv[200, c:2 * c] = qkv[3, 0:c] * 6.0  # key 200 aligned with query 3
    qkv[70, c:2 * c] = qkv[9, 0:c] * 4.0
    qkv = qkv.to(dtype)
    q, kk, v = [t.float().reshape(images, tokens, heads, d).transpose(1, 2) for t in qkv.split(c, dim=1)]
    ref = _sdpa(q, kk, v).transpose(1, 2).reshape(images * tokens, c)
    out = k.attention_spatial(qkv.to(DEV), images, tokens, heads)
    torch.cuda.synchronize()
    close(out, ref, dtype, "attn rescale", rel=8e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("b,f,tokens,heads,d,L,strip", [(2, 4, 64, 8, 40, 77, 0), (1, 3, 50, 8, 80, 81, 4), (2, 2, 16, 8, 160, 77, 0),
                                                    # (>= 256 queries, 65..80 keys, 8 heads of 40 / 80: the register-resident K/V kernel k_attn_short)
                                                    (2, 3, 300, 8, 40, 77, 0), (1, 3, 272, 8, 80, 81, 4), (3, 2, 1000, 8, 80, 70, 0), (2, 16, 256, 8, 40, 80, 0)])
def test_attention_cross_and_ip(b, f, tokens, heads, d, L, strip, dtype):
    k = _k()
    c = heads * d
    images = b * f
    q = rnd(images * tokens, c, dtype=dtype, seed=43)
    kv = rnd(b * L, 2 * c, dtype=dtype, seed=44)
    nk = L - strip
    qh = q.float().reshape(images, tokens, heads, d).transpose(1, 2)
    kh = kv[:, :c].float().reshape(b, L, heads, d)[:, :nk].transpose(1, 2).repeat_interleave(f, 0)
    vh = kv[:, c:].float().reshape(b, L, heads, d)[:, :nk].transpose(1, 2).repeat_interleave(f, 0)
    ref = _sdpa(qh, kh, vh).transpose(1, 2).reshape(images * tokens, c)
    out = k.attention_cross(q.to(DEV), kv.to(DEV), images, tokens, heads, nk, L, f)
    torch.cuda.synchronize()
    close(out, ref, dtype, "attn_cross", rel=8e-3 if dtype == torch.bfloat16 else 3e-3)
    if strip:
        # IP-Adapter: second attention over the last `strip` rows of an ip K/V buffer, accumulated with a scale
        kvip = rnd(b * strip, 2 * c, dtype=dtype, seed=45)
        kh2 = kvip[:, :c].float().reshape(b, strip, heads, d).transpose(1, 2).repeat_interleave(f, 0)
        vh2 = kvip[:, c:].float().reshape(b, strip, heads, d).transpose(1, 2).repeat_interleave(f, 0)
        ref2 = out.float().cpu() + 0.4 * _sdpa(qh, kh2, vh2).transpose(1, 2).reshape(images * tokens, c)
        k.attention_cross(q.to(DEV), kvip.to(DEV), images, tokens, heads, strip, strip, f, out=out, out_scale=0.4,
                          accumulate=True)
        torch.cuda.synchronize()
        close(out, ref2, dtype, "attn_ip_accumulate", rel=1e-2 if dtype == torch.bfloat16 else 4e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("b,f,tokens,heads,d", [(2, 16, 20, 8, 40), (1, 8, 64, 8, 80), (2, 24, 9, 8, 160), (1, 32, 5, 8, 16), (2, 16, 256, 8, 40)])
def test_attention_temporal(b, f, tokens, heads, d, dtype):
    k = _k()
    c = heads * d
    qkv = rnd(b * f * tokens, 3 * c, dtype=dtype, seed=46)
    # rows (b f n) -> per (b, n): sequence over f
    t = qkv.float().reshape(b, f, tokens, 3, heads, d).permute(3, 0, 2, 4, 1, 5)  # [3, b, n, h, f, d]
    ref = _sdpa(t[0], t[1], t[2])  # [b, n, h, f, d]
    ref = ref.permute(0, 3, 1, 2, 4).reshape(b * f * tokens, c)
    out = k.attention_temporal(qkv.to(DEV), b, f, tokens, heads)
    torch.cuda.synchronize()
    close(out, ref, dtype, f"attn_temporal({b},{f},{tokens},{heads},{d})", rel=8e-3 if dtype == torch.bfloat16 else 3e-3)


# ------------------------------------------------------------------------------------ elementwise
def test_add_bcast():
    k = _k()
    a = rnd(2, 4, 8, 8, 64, seed=51)
    b = rnd(1, 4, 8, 8, 64, seed=52)
    out = k.add_bcast(a.to(DEV), b.to(DEV))
    torch.cuda.synchronize()
    close(out, a.float() + b.float(), torch.bfloat16, "add_bcast")


def test_repeat_batch_equals_torch_cat():
    """ca_repeat (ABI v8): the CFG halves of the shared prefix, torch.cat([x, x]) with one read."""
    k = _k()
    for shape, dt in (((16, 64, 64, 320), torch.float16), ((3, 7, 8), torch.bfloat16), ((5, 4), torch.float32)):
        x = rnd(*shape, dtype=dt, seed=3).to(DEV)
        for times in (2, 3):
            assert torch.equal(k.repeat_batch(x, times), torch.cat([x] * times))
    # sizes / alignments / strides the 16-byte kernel cannot take (round-3 advice: these used to assert)
    x = rnd(7, 3, 5, dtype=torch.float16, seed=4).to(DEV)                   # 210 bytes
    assert torch.equal(k.repeat_batch(x), torch.cat([x, x]))
    base = rnd(9, 8, 8, dtype=torch.float16, seed=5).to(DEV)
    v = base[1:5]                                                          # contiguous view with a storage offset
    assert torch.equal(k.repeat_batch(v, 3), torch.cat([v] * 3))
    v = base.view(-1)[1:1 + 64].view(4, 16)                                 # 2-byte aligned start
    assert v.data_ptr() % 16 != 0 and torch.equal(k.repeat_batch(v), torch.cat([v, v]))
    v = base[:, :, ::2]                                                    # strided
    assert torch.equal(k.repeat_batch(v), torch.cat([v, v]))


def test_silu_and_timestep_embedding():
    k = _k()
    x = rnd(2, 1280, dtype=torch.float32, seed=53)
    y = k.silu_f32(x.to(DEV))
    torch.cuda.synchronize()
    assert torch.allclose(y.cpu(), F.silu(x), atol=1e-5, rtol=1e-5)
    half = 160
    freq = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    for t in (999.0, 3.0):
        emb = k.timestep_embedding(t, 2, 320, torch.bfloat16, DEV)
        arg = t * freq
        ref = torch.cat([torch.cos(arg), torch.sin(arg)])[None].repeat(2, 1)
        torch.cuda.synchronize()
        assert (emb.float().cpu() - ref).abs().max().item() < 6e-3
    tt = torch.tensor([10.0, 500.0], device=DEV)
    emb = k.timestep_embedding(tt, 2, 320, torch.float16, DEV)
    ref = torch.stack([torch.cat([torch.cos(t * freq), torch.sin(t * freq)]) for t in (10.0, 500.0)])
    torch.cuda.synchronize()
    assert (emb.float().cpu() - ref).abs().max().item() < 2e-3


def test_layout_kernels_and_scheduler_step():
    k = _k()
    lat = rnd(1, 4, 6, 8, 10, dtype=torch.float32, seed=54)
    nh = k.latents_to_nhwc(lat.to(DEV), 8, 2, 0.5, torch.bfloat16)
    torch.cuda.synchronize()
    ref = (lat * 0.5).permute(0, 2, 3, 4, 1).reshape(6, 8, 10, 4)
    ref = torch.cat([ref, torch.zeros(6, 8, 10, 4)], -1).repeat(2, 1, 1, 1)
    close(nh, ref, torch.bfloat16, "latents_to_nhwc")
    back = k.nhwc_to_ncfhw_f32(nh, 2, 4, 6)
    torch.cuda.synchronize()
    close(back[0:1], lat * 0.5, torch.bfloat16, "nhwc_to_ncfhw")
    x5 = rnd(2, 320, 3, 4, 5, dtype=torch.float16, seed=55).to(DEV)
    xv = x5.permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3)  # channels-last strides, 5-D logical
    for src in (x5, xv, x5.float()):
        o = k.ncfhw_to_nhwc(src, 320, torch.bfloat16)
        torch.cuda.synchronize()
        close(o, x5.float().cpu().permute(0, 2, 3, 4, 1).reshape(6, 4, 5, 320), torch.bfloat16, "ncfhw_to_nhwc")
    # scheduler step
    f, h, w = 6, 8, 10
    eps = rnd(2 * f, h, w, 4, dtype=torch.float32, seed=56)
    noise = rnd(1, 4, f, h, w, dtype=torch.float32, seed=57)
    coef = [0.3, 1.7, 0.9, 0.1, 0.8, 0.2, 0.5]
    g = 7.5
    prev, den = k.cfg_scheduler_step(eps.to(DEV), 2, g, lat.to(DEV), noise.to(DEV), coef, clip=1.0, want_denoised=True)
    torch.cuda.synchronize()
    e = eps.reshape(2, f, h, w, 4).permute(0, 4, 1, 2, 3)
    e = e[0:1] + g * (e[1:2] - e[0:1])
    x0 = ((lat - coef[0] * e) * coef[1]).clamp(-1, 1)
    dref = coef[2] * x0 + coef[3] * lat
    pref = coef[4] * dref + coef[5] * e + coef[6] * noise
    assert torch.allclose(den.cpu(), dref, atol=1e-5, rtol=1e-5)
    assert torch.allclose(prev.cpu(), pref, atol=1e-5, rtol=1e-5)


def test_lincomb_and_cfg_combined_eps():
    """ca_lincomb (update rule of the history-carrying samplers) and the CFG combine alone."""
    from controlanimate_amd import kernels as K
    g = torch.Generator().manual_seed(3)
    xs = [torch.randn(1, 4, 5, 6, 7, generator=g).to(DEV) for _ in range(5)]
    cs = [0.7, -1.3, 2.5, 1e-3, -0.25]
    out = K.lincomb(list(zip(xs, cs)))
    ref = sum(c * x.double() for x, c in zip(xs, cs))
    assert torch.allclose(out.double(), ref, atol=1e-5)
    inplace = K.lincomb([(xs[0], 2.0), (xs[1], 1.0)], out=xs[0].clone())
    assert torch.allclose(inplace, 2.0 * xs[0] + xs[1], atol=1e-6)
    f, h, w = 5, 6, 7
    eps = torch.randn(2 * f, h, w, 4, generator=g).to(DEV)
    lat = torch.randn(1, 4, f, h, w, generator=g).to(DEV)
    e = K.cfg_combined_eps(eps, 2, 7.5, lat)
    eu, ec = eps[:f].permute(3, 0, 1, 2)[None], eps[f:].permute(3, 0, 1, 2)[None]
    assert torch.allclose(e, eu + 7.5 * (ec - eu), atol=1e-5)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_weight_resident_k320(dtype):
    """The weight-resident streaming kernel (ca_gemm_wres.h; every K = 320 GEMM with M >= 16384 -- the 64x64-latent
    level -- takes it): every epilogue it implements, ragged M, strided operands, both wave roles of the row-bias patch,
    folded-LayerNorm statistics computed inside the kernel (ABI v6) against the separate statistics pass;
    each case against an fp32 reference with the tiled kernels' rounding order, and bit-for-bit repeatable."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import wres_check
    k = _k()
    tol = 2e-3 if dtype == torch.float16 else 1.2e-2
    for name, kw in wres_check.cases(torch, "cuda", dtype, small=True):
        outs = [wres_check.call(k, kw).clone() for _ in range(2)]
        ref = wres_check.reference(torch, F, kw)
        rel = ((outs[0].float() - ref).norm() / ref.norm()).item()
        assert torch.isfinite(outs[0].float()).all() and rel < tol, (name, rel)
        assert torch.equal(outs[0], outs[1]), name


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_activation_resident_k320(dtype):
    """The activation-resident kernel (ca_gemm_ar.h, ABI v9: `w_frag`): q|k|v and GEGLU projections of the 64x64-latent level.
    Every epilogue it implements (bias, residual, row bias per frame, folded LayerNorm with finished or in-kernel statistics,
    GEGLU), ragged M (a last tile of 8 / 72 rows), strided A / C -- each against an fp32 reference, bit-for-bit repeatable,
    dispatched to the kernel this test is about, and the same call WITHOUT the fragment-ordered weights (the weight-resident
    kernel) within rounding of it.  ca_pack_w_frag == layers.frag_order (what the arena holds)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import ar_check
    from controlanimate_amd.layers import frag_order
    k = _k()
    tol = 2e-3 if dtype == torch.float16 else 1.2e-2
    for name, kw in ar_check.cases(dtype, small=True):
        lab = ar_check.label(lambda: ar_check.call(kw, True))
        assert lab == "ar128x64", (name, lab)
        outs = [ar_check.call(kw, True).clone() for _ in range(2)]
        old = ar_check.call(kw, False)
        ref = ar_check.reference(kw)
        rel = ((outs[0].float() - ref).norm() / ref.norm()).item()
        assert torch.isfinite(outs[0].float()).all() and rel < tol, (name, rel)
        assert torch.equal(outs[0], outs[1]), name
        assert ((outs[0].float() - old.float()).norm() / ref.norm()).item() < tol / 4, name
        if "ln" not in kw and not kw.get("geglu"):  # (the statistics' summation order differs between the kernels; GEGLU products a last bit)
            assert torch.equal(outs[0], old), name
    for geglu in (False, True):
        w = rnd(2560, 320, dtype=dtype, seed=11).to(DEV)
        k.attach_w_frag(w, geglu)
        assert torch.equal(w._frag[0], frag_order(w, geglu)) and w._frag[1] == geglu
    # a weight the kernel cannot take gets no twin
    w = rnd(640, 640, dtype=dtype, seed=12).to(DEV)
    assert not hasattr(k.attach_w_frag(w), "_frag")


@pytest.mark.parametrize("dtype", DTYPES)
def test_ff_fused_c320(dtype):
    """ca_ff_fused (ABI v9, csrc/ca_ff_fused.h): the feed-forward of the 64x64-latent level -- folded LayerNorm with in-kernel
    statistics, GEGLU projection, output projection, residual -- in one launch, against fp32 torch (the reference's arithmetic:
    animatediff/models/attention.py:288,303-357) and against the two ca_gemm calls it replaces; ragged M (a last tile of 72 rows),
    strided x, no residual; bit-for-bit repeatable; ca_pack_w2_frag == layers.frag_order2; outside its shapes the entry says no."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import ff_check
    from controlanimate_amd import _capi
    from controlanimate_amd.layers import frag_order2
    k = _k()
    tol = 2e-3 if dtype == torch.float16 else 1.2e-2
    for (m, lda, res) in ((16384, 320, True), (16384 + 72, 320, True), (20480, 640, False)):
        d = ff_check.make(m, dtype, lda=lda)
        k._plan_sink = labels = []
        try:
            outs = [ff_check.fused(d, res) for _ in range(2)]
        finally:
            k._plan_sink = None
        assert outs[0] is not None and labels == ["ff_fused128", "ff_fused128"]
        ref, two = ff_check.reference(d, res), ff_check.two_gemms(d, res)
        rel = ((outs[0].float() - ref).norm() / ref.norm()).item()
        assert torch.isfinite(outs[0].float()).all() and rel < tol, (m, rel)
        assert torch.equal(outs[0], outs[1])
        # (the fused kernel normalises its x tile in place, i.e. rounds LN(x) to the activation type as the reference does; the
        #  two-GEMM path folds (mean, rstd) into the epilogue in fp32: one operand rounding apart)
        assert ((outs[0].float() - two.float()).norm() / ref.norm()).item() < tol / 2
    w2 = rnd(320, 1280, dtype=dtype, seed=13).to(DEV)
    dst = torch.empty_like(w2)
    _capi.check(k.lib().ca_pack_w2_frag(w2.data_ptr(), 320, 1280, dst.data_ptr(), None), "ca_pack_w2_frag")
    torch.cuda.synchronize()
    assert torch.equal(dst, frag_order2(w2))
    # shapes the kernel does not take: the caller gets None and runs the two GEMMs
    d = ff_check.make(8192, dtype)
    assert ff_check.fused(d) is None


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("m,c", [(8192, 1280), (32768 - 24, 640)])
def test_layernorm_statistics_handed_from_producer_to_consumer(m, c, dtype):
    """ABI v6 row_sums_out / ln_parts: the GEMM that writes a tensor leaves (sum, sum of squares) of every stored row per
    320-column tile; the GEMM with the folded LayerNorm of that tensor finishes mean / rstd from them -- same result as
    with the separate statistics pass, which is skipped."""
    k = _k()
    a = rnd(m, c, dtype=dtype, seed=71).to(DEV)
    w = rnd(c, c, dtype=torch.float32, scale=c ** -0.5, seed=72).to(dtype).to(DEV)
    res = (rnd(m, c, dtype=torch.float32, seed=73) * 1.5 + 0.7).to(dtype).to(DEV)
    x = k.gemm(a, w, residual=res, row_sums=True)
    sums = k.row_sums_of(x)
    assert sums is not None, "the 128x320-tile kernel takes this shape: row sums expected"
    rs, parts = sums
    # one partial sum per 320-column tile (128x320-tile kernels) or per 80-column wave quarter (256x320 kernel, ABI v8)
    assert parts in (c // 320, 4 * (c // 320)) and rs.shape == (m, parts, 2)
    xf = x.float().view(m, parts, c // parts)
    assert torch.allclose(rs[..., 0], xf.sum(-1), rtol=1e-4, atol=2e-2) and torch.allclose(rs[..., 1], (xf * xf).sum(-1), rtol=1e-4, atol=2e-2)
    n = 3 * c
    w2 = rnd(n, c, dtype=torch.float32, scale=c ** -0.5, seed=74).to(dtype).to(DEV)
    cs = w2.float().sum(1).contiguous()
    bias = rnd(n, dtype=torch.float32, seed=75).to(DEV)
    y_sep = k.gemm(x, w2, bias=bias, ln=(k.row_stats(x, 1e-5), cs))
    y_new = k.gemm(x, w2, bias=bias, ln=(k.RowStats(x, 1e-5, sums), cs))
    torch.cuda.synchronize()
    ref = F.linear(F.layer_norm(x.float(), (c,), eps=1e-5), w2.float(), bias)
    close(y_new, ref.cpu(), dtype, f"ln via producer sums ({m},{c})")
    assert ((y_new.float() - y_sep.float()).abs().max() <= 4 * 2.0 ** (-8 if dtype == torch.bfloat16 else -10) * ref.abs().max()).item()
    # a shape the 128x320 kernel does not take: no sums, and asking for them in the C call is an error, not a fallback
    small = k.gemm(a[:256], w, row_sums=True)
    assert k.row_sums_of(small) is None


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_and_conv_256x320_streaming_kernel(dtype):
    """ca_gemm_pq.h (256 x 320 tiles, 128 x 80 wave tiles; round 3): the launches the plan sends to it by default -- long-K
    dense GEMMs and the 32x32-latent convolutions with one tile per CU -- with every epilogue it implements (bias, one and
    two row-bias groups per tile, alpha, residual), ragged M, two-source and stride-2 gathers; each against an fp32
    reference with the tiled kernels' rounding order, bit-for-bit repeatable, and on the kernel the case is about."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import ps_check
    k = _k()
    tol = 2.5e-3 if dtype == torch.float16 else 1.5e-2
    g = torch.Generator().manual_seed(5)

    def rn(*s, scale=1.0):
        return (torch.randn(*s, generator=g) * scale).to(DEV)

    cases = []
    for (m, n, kk) in [(32768, 640, 2560), (32768 - 40, 640, 640), (65536, 320, 1280)]:
        a, w = rn(m, kk).to(dtype), rn(n, kk, scale=kk ** -0.5).to(dtype)
        bias, res = rn(n), rn(m, n).to(dtype)
        cases.append(("gemm", f"plain {m}x{n}x{kk}", dict(a=a, w=w)))
        cases.append(("gemm", f"bias+res {m}x{n}x{kk}", dict(a=a, w=w, bias=bias, residual=res)))
        cases.append(("gemm", f"rowbias(128)+res alpha {m}x{n}x{kk}", dict(a=a, w=w, bias=bias, rowbias=rn((m + 127) // 128, n), rows_per_group=128, residual=res, alpha=0.75)))
        cases.append(("gemm", f"rowbias(1024) {m}x{n}x{kk}", dict(a=a, w=w, rowbias=rn((m + 1023) // 1024, n), rows_per_group=1024)))
    for (img, h, ci, co, stride, c2) in [(32, 32, 640, 640, 1, 0), (32, 32, 640, 640, 1, 640), (37, 30, 320, 640, 1, 0), (32, 64, 320, 640, 2, 0)]:
        x = rn(img, h, h, ci).to(dtype)
        x2 = rn(img, h, h, c2).to(dtype) if c2 else None
        w = rn(co, 3, 3, ci + c2, scale=(9 * (ci + c2)) ** -0.5).to(dtype)
        ho = (h + 2 - 3) // stride + 1
        cases.append(("conv", f"conv {img}x{h} {ci}+{c2}->{co} s{stride}", dict(x=x, x2=x2, w=w, stride=stride)))
        full = dict(x=x, x2=x2, w=w, stride=stride, bias=rn(co), residual=rn(img, ho, ho, co).to(dtype))
        if (ho * ho) % 128 == 0:  # (row-bias groups = images; the kernel takes multiples of 128 rows)
            full.update(rowbias=rn(img, co), rows_per_group=ho * ho)
        cases.append(("conv", f"conv+bias+rowbias+res {img}x{h} {ci}+{c2}->{co} s{stride}", full))
    # folded LayerNorm (+ GEGLU): finished statistics, and the producer's partial sums (finished by ca_ln_finish_sums, ABI v8)
    for (m, n, kk, geglu) in [(32768 - 24, 1920, 640, False), (8192, 10240, 1280, True), (2048, 10240, 1280, True)]:
        a, w = (rn(m, kk) * 1.3 + 0.4).to(dtype), rn(n, kk, scale=kk ** -0.5).to(dtype)
        bias = rn(n)
        st = torch.stack([a.float().mean(1), (a.float().var(1, unbiased=False) + 1e-5).rsqrt()], 1).contiguous()
        cs = w.float().sum(1).contiguous()
        cases.append(("gemm", f"LN (mean, rstd){' geglu' if geglu else ''} {m}x{n}x{kk}", dict(a=a, w=w, bias=bias, geglu=geglu, ln=(st, cs), _ln_ref=st)))
        parts = kk // 320
        af = a.float().reshape(m, parts, 320)
        sums = torch.stack([af.sum(2), (af * af).sum(2)], 2).contiguous()
        cases.append(("gemm", f"LN {parts} partial sums{' geglu' if geglu else ''} {m}x{n}x{kk}",
                      dict(a=a, w=w, bias=bias, geglu=geglu, ln=(k.RowStats(a, 1e-5, (sums, parts)), cs), _ln_ref=st)))
    for kind, name, kw in cases:
        k._plan_sink = labels = []
        if kind == "conv":
            call = dict(kw)
            x, w = call.pop("x"), call.pop("w")
            outs = [k.conv3x3(x, w, **call).clone() for _ in range(2)]
            ref = ps_check.conv_reference(kw)
        else:
            outs = [k.gemm(**{a_: b_ for a_, b_ in kw.items() if not a_.startswith("_")}).clone() for _ in range(2)]
            ref = ps_check.gemm_reference(kw)
        k._plan_sink = None
        assert set(labels) == {"pq256x320"}, (name, labels)
        rel = ((outs[0].float() - ref).norm() / ref.norm()).item()
        assert torch.isfinite(outs[0].float()).all() and rel < tol, (name, rel)
        assert torch.equal(outs[0], outs[1]), name


@pytest.mark.gpu
@pytest.mark.parametrize("dt,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.2e-2)])
def test_tattn_fused_c320(dt, tol):
    """ca_tattn_fused (ABI v10): LayerNorm + positional encoding + q|k|v + attention over the 16 frames in one launch, against fp32
    torch and against the two-launch path it replaces; the library's weight packing against layers.frag_order_tattn."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import tattn_check as T
    from controlanimate_amd import kernels as K
    from controlanimate_amd.layers import frag_order_tattn
    for (b, tokens, lda) in [(2, 1024, 320), (1, 1032, 640)]:
        x, w, gamma, beta, pe = T.make(b, tokens, dt, lda=lda)
        ref = T.reference(x, w, gamma, beta, pe, b, tokens)
        wl = torch.empty(368640, device="cuda", dtype=dt)
        K.check(K.lib().ca_pack_w_tattn(w.data_ptr(), 960, 320, wl.data_ptr(), K._stream()), "ca_pack_w_tattn")
        assert torch.equal(wl, frag_order_tattn(w.float()).to(dt))
        o = T.fused(x, w, gamma, beta, pe, b, tokens, wl)
        assert o is not None, "the library must take this shape"
        assert torch.equal(o, T.fused(x, w, gamma, beta, pe, b, tokens, wl))
        rel = ((o.float() - ref).norm() / ref.norm()).item()
        old = T.two_launch(x, w, gamma, beta, pe, b, tokens)
        rel_old = ((old.float() - ref).norm() / ref.norm()).item()
        assert rel < tol and rel < 1.5 * rel_old + 1e-4, (rel, rel_old)
    # shapes the kernel does not implement are declined, not mis-run
    x, w, gamma, beta, pe = T.make(1, 64, dt)   # 1024 rows: too few
    assert T.fused(x, w, gamma, beta, pe, 1, 64) is None


@pytest.mark.gpu
@pytest.mark.parametrize("dt,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.2e-2)])
def test_tattn_fused_with_output_projection_c320(dt, tol):
    """ca_tattn_fused with w_out_frag (ABI v12): the attention, its output projection, bias and residual in one launch -- against fp32
    torch (motion_module.py:212-224: `attention_block(norm(x)) + x`) and against the two launches it replaces (ca_tattn_fused + ca_gemm);
    the library's packing against layers.frag_order_wout; bit-reproducible."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import tattn_check as T
    from controlanimate_amd import kernels as K
    from controlanimate_amd.layers import frag_order_wout
    for (b, tokens, lda, res) in [(2, 1024, 320, True), (1, 1032, 640, True), (2, 4096, 320, True), (1, 1024, 320, False)]:
        x, w, gamma, beta, pe = T.make(b, tokens, dt, lda=lda)
        wo, bo = T.make_out(dt)
        ref = T.reference(x, w, gamma, beta, pe, b, tokens) @ wo.float().t() + bo[None, :] + (x.float() if res else 0.0)
        wol = torch.empty(102400, device="cuda", dtype=dt)
        K.check(K.lib().ca_pack_w_out(wo.data_ptr(), 320, 320, wol.data_ptr(), K._stream()), "ca_pack_w_out")
        assert torch.equal(wol, frag_order_wout(wo.float()).to(dt))
        K._plan_sink = labels = []
        try:
            y = T.fused_out(x, w, gamma, beta, pe, b, tokens, wo, bo, wofrag=wol, residual=res)
        finally:
            K._plan_sink = None
        assert y is not None and labels == ["tattn_out128"], "the library must take this shape"
        assert torch.equal(y, T.fused_out(x, w, gamma, beta, pe, b, tokens, wo, bo, wofrag=wol, residual=res))
        rel = ((y.float() - ref).norm() / ref.norm()).item()
        old = T.two_launch_out(x, w, gamma, beta, pe, b, tokens, wo, bo, residual=res)
        rel_old = ((old.float() - ref).norm() / ref.norm()).item()
        assert rel < tol and rel < 1.5 * rel_old + 1e-4, (rel, rel_old)
    # a bias or a residual without the projection is refused, not silently dropped
    x, w, gamma, beta, pe = T.make(2, 1024, dt)
    bp = (pe + beta[None, :]).contiguous()
    from controlanimate_amd.layers import frag_order_tattn
    assert K.tattn_fused(x, frag_order_tattn(w.float()).to(dt), gamma.contiguous(), bp, 2, 16, 1024, 8, 1e-5, 40 ** -0.5, residual=x) is None


@pytest.mark.gpu
@pytest.mark.parametrize("dt,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.2e-2)])
def test_xattn_fused_c320(dt, tol):
    """ca_xattn_fused (ABI v11): LayerNorm + to_q + attention over the text tokens in one launch, against fp32 torch and against the
    two launches it replaces; the library's packing kernels against layers.frag_order_xattn; K / V fragments repacked in place."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import xattn_check as X
    from controlanimate_amd import kernels as K
    from controlanimate_amd.layers import frag_order_xattn
    for (images, tokens, L, nk, fpk, kvb, lda) in [(8, 2048, 77, 77, 4, 2, 320), (6, 3072, 81, 77, 2, 3, 640), (16, 1024, 70, 70, 8, 2, 320)]:
        x, wq, gamma, beta, kv = X.make(images, tokens, L, kvb, dt, lda=lda)
        ref = X.reference(x, wq, gamma, beta, kv, images, tokens, L, nk, fpk, kvb, dt)
        wf, cs, bias = X.operands(x, wq, gamma, beta, dt)
        wl = torch.empty(122880, device="cuda", dtype=dt)
        K.check(K.lib().ca_xattn_pack_w(wf.data_ptr(), 320, 320, wl.data_ptr(), K._stream()), "ca_xattn_pack_w")
        assert torch.equal(wl, frag_order_xattn(wf.float()).to(dt))
        kvf = K.xattn_pack_kv(kv, kvb, L, nk, 40 ** -0.5)
        o = X.fused(x, wq, gamma, beta, kv, images, tokens, L, nk, fpk, kvb, (wl, kvf))
        assert o is not None, "the library must take this shape"
        assert torch.equal(o, X.fused(x, wq, gamma, beta, kv, images, tokens, L, nk, fpk, kvb, (wl, kvf)))
        rel = ((o.float() - ref).norm() / ref.norm()).item()
        old = X.two_launch(x, wq, gamma, beta, kv, images, tokens, L, nk, fpk, kvb)
        rel_old = ((old.float() - ref).norm() / ref.norm()).item()
        assert rel < tol and rel < 1.5 * rel_old + 1e-4, (rel, rel_old)
        # a new window: other K / V in the same buffers, fragments repacked in place
        kv2 = X.make(images, tokens, L, kvb, dt, seed=9, lda=lda)[4]
        kv.copy_(kv2)
        assert K.xattn_pack_kv(kv, kvb, L, nk, 40 ** -0.5, out=kvf) is kvf
        o2 = X.fused(x, wq, gamma, beta, kv, images, tokens, L, nk, fpk, kvb, (wl, kvf))
        ref2 = X.reference(x, wq, gamma, beta, kv, images, tokens, L, nk, fpk, kvb, dt)
        assert ((o2.float() - ref2).norm() / ref2.norm()).item() < tol
    # declined, not mis-run: too few rows; 64 keys
    x, wq, gamma, beta, kv = X.make(2, 1024, 77, 1, dt)
    assert X.fused(x, wq, gamma, beta, kv, 2, 1024, 77, 77, 1, 1) is None
    assert K.xattn_pack_kv(kv, 1, 77, 64, 40 ** -0.5) is None


@pytest.mark.gpu
@pytest.mark.parametrize("dt,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.2e-2)])
def test_xattn_fused_with_output_projection_c320(dt, tol):
    """ca_xattn_fused with w_out_frag (ABI v12): text cross-attention + to_out + bias + residual in one launch -- against fp32 torch
    (animatediff/models/attention.py:253-262, modules/attention_processor.py:258-270) and against the two launches it replaces."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import xattn_check as X
    from controlanimate_amd import kernels as K
    for (images, tokens, L, nk, fpk, kvb, lda, res) in [(8, 2048, 77, 77, 4, 2, 320, True), (6, 3072, 81, 77, 2, 3, 640, True), (16, 1024, 70, 70, 8, 2, 320, False)]:
        x, wq, gamma, beta, kv = X.make(images, tokens, L, kvb, dt, lda=lda)
        wo, bo = X.make_out(dt)
        ref = X.reference(x, wq, gamma, beta, kv, images, tokens, L, nk, fpk, kvb, dt) @ wo.float().t() + bo[None, :] + (x.float() if res else 0.0)
        K._plan_sink = labels = []
        try:
            y = X.fused_out(x, wq, gamma, beta, kv, images, tokens, L, nk, fpk, kvb, wo, bo, residual=res)
        finally:
            K._plan_sink = None
        assert y is not None and labels == ["xattn_out128"], "the library must take this shape"
        assert torch.equal(y, X.fused_out(x, wq, gamma, beta, kv, images, tokens, L, nk, fpk, kvb, wo, bo, residual=res))
        rel = ((y.float() - ref).norm() / ref.norm()).item()
        old = X.two_launch_out(x, wq, gamma, beta, kv, images, tokens, L, nk, fpk, kvb, wo, bo, residual=res)
        rel_old = ((old.float() - ref).norm() / ref.norm()).item()
        assert rel < tol and rel < 1.5 * rel_old + 1e-4, (rel, rel_old)


@pytest.mark.gpu
@pytest.mark.parametrize("dt,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.2e-2)])
def test_xattn_fused_with_image_prompt_tokens_c320(dt, tol):
    """ca_xattn_fused with kv_frag_ip (ABI v13): the IP-Adapter's cross-attention site -- attention over the text tokens, a second attention
    of the same q over the image-prompt tokens (the last num_tokens context rows through to_k_ip / to_v_ip), `o + scale * o_ip`, to_out +
    bias + residual -- in one launch, against fp32 torch (modules/attention_processor.py:433-477) and the four launches it replaces."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import xattn_check as X
    from controlanimate_amd import kernels as K
    for (images, tokens, L, nk, nip, fpk, kvb, lda, res, sc) in [(8, 2048, 81, 77, 4, 4, 2, 320, True, 1.0), (6, 3072, 88, 72, 16, 2, 3, 640, True, 0.6),
                                                                 (16, 1024, 78, 77, 1, 8, 2, 320, False, 0.5)]:
        x, wq, gamma, beta, kv = X.make(images, tokens, L, kvb, dt, lda=lda)
        kvip = X.make(images, tokens, L, kvb, dt, seed=23, lda=lda)[4]
        wo, bo = X.make_out(dt)
        ref = (X.reference_ip(x, wq, gamma, beta, kv, kvip, images, tokens, L, nk, nip, fpk, kvb, dt, sc) @ wo.float().t() + bo[None, :]
               + (x.float() if res else 0.0))
        K._plan_sink = labels = []
        try:
            y = X.fused_ip_out(x, wq, gamma, beta, kv, kvip, images, tokens, L, nk, nip, fpk, kvb, wo, bo, sc, residual=res)
        finally:
            K._plan_sink = None
        assert y is not None and labels == ["xattn_ip_out128"], "the library must take this shape"
        assert torch.equal(y, X.fused_ip_out(x, wq, gamma, beta, kv, kvip, images, tokens, L, nk, nip, fpk, kvb, wo, bo, sc, residual=res))
        rel = ((y.float() - ref).norm() / ref.norm()).item()
        old = X.separate_ip_out(x, wq, gamma, beta, kv, kvip, images, tokens, L, nk, nip, fpk, kvb, wo, bo, sc, residual=res)
        rel_old = ((old.float() - ref).norm() / ref.norm()).item()
        assert torch.isfinite(y.float()).all() and rel < tol and rel < 1.5 * rel_old + 1e-4, (rel, rel_old)
        # the image-prompt tokens matter: without them the result is a different one
        y0 = X.fused_out(x, wq, gamma, beta, kv, images, tokens, L, nk, fpk, kvb, wo, bo, residual=res)
        assert ((y0.float() - ref).norm() / ref.norm()).item() > 5 * rel
    # the image-prompt set needs the output stage; more than 16 tokens are not taken
    x, wq, gamma, beta, kv = X.make(8, 2048, 81, 2, dt)
    wf, cs, bias = X.operands(x, wq, gamma, beta, dt)
    from controlanimate_amd.layers import frag_order_xattn
    kvf, kvf_ip = K.xattn_pack_kv(kv, 2, 81, 77, 40 ** -0.5), K.xattn_pack_kv(kv, 2, 81, 4, 40 ** -0.5, row_offset=77)
    assert K.xattn_fused(x, frag_order_xattn(wf.float()).to(dt), bias, kvf, 8, 2048, 4, 2, 77, 1e-5, kv_frag_ip=kvf_ip, nk_ip=4) is None
    assert K.xattn_pack_kv(kv, 2, 81, 17, 40 ** -0.5, row_offset=64) is None


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", DTYPES)
def test_ff_fused_with_proj_out_c320(dtype):
    """ca_ff_fused with w_out_frag (ABI v12): the feed-forward, the transformer's proj_out, its bias and the transformer's residual in one
    launch (animatediff/models/attention.py:163-175, motion_module.py:158-163) -- against fp32 torch and the two launches it replaces."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import ff_check
    k = _k()
    tol = 2e-3 if dtype == torch.float16 else 1.2e-2
    for (m, lda, res, with_res_out) in ((16384, 320, True, True), (32768, 640, True, True), (16384, 320, False, False)):
        d = ff_check.make(m, dtype, lda=lda)
        wo, bo = ff_check.make_out(dtype)
        res_out = rnd(m, 320, dtype=dtype, seed=31).to(DEV) if with_res_out else None
        k._plan_sink = labels = []
        try:
            outs = [ff_check.fused_out(d, wo, bo, res_out, res) for _ in range(2)]
        finally:
            k._plan_sink = None
        assert outs[0] is not None and labels == ["ff_out128", "ff_out128"]
        ref = ff_check.reference(d, res) @ wo.float().t() + bo[None, :] + (res_out.float() if with_res_out else 0.0)
        two = ff_check.two_launch_out(d, wo, bo, res_out, res)
        rel = ((outs[0].float() - ref).norm() / ref.norm()).item()
        rel_two = ((two.float() - ref).norm() / ref.norm()).item()
        assert torch.isfinite(outs[0].float()).all() and rel < tol and rel < 1.5 * rel_two + 1e-4, (m, rel, rel_two)
        assert torch.equal(outs[0], outs[1])
    # the output stage works on whole row tiles: a ragged M is declined (the caller runs the feed-forward, then proj_out)
    d = ff_check.make(16384 + 72, dtype)
    wo, bo = ff_check.make_out(dtype)
    assert ff_check.fused_out(d, wo, bo, None) is None


@pytest.mark.gpu
def test_conv3x3_winograd_large_activations_stay_finite():
    """The Winograd route keeps the transformed input V (sums of four activations) and the sixteen transformed products M in the
    activation type, so its overflow headroom in fp16 is smaller than the direct form's.  Large-magnitude operands -- inputs
    of std 24 (max ~ +-120, V up to 4x that), weights 4x the usual scale, outputs of std ~ 100 with a residual of the same size -- must
    stay finite and within the direct form's distance of fp32 torch."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import wino_check as W
    k = _k()
    for (images, h, c1, c2, cout) in [(32, 16, 1280, 1280, 1280), (32, 8, 1280, 0, 1280)]:
        d = W.make(images, h, c1, c2, cout, epilogue=True, seed=11)
        d["x"] = (d["x"].float() * 24.0).half()
        if d["x2"] is not None:
            d["x2"] = (d["x2"].float() * 24.0).half()
        d["w32"] = d["w32"] * 4.0
        d["w"] = d["w32"].permute(0, 2, 3, 1).contiguous().half()
        d["u"] = torch.einsum("xk,oikl,yl->xyoi", W.G.to(DEV), d["w32"], W.G.to(DEV)).reshape(16, cout, c1 + c2).contiguous().half()
        d["residual"] = (d["residual"].float() * 100.0).half()
        k._plan_sink = labels = []
        try:
            y, direct = W.run(d, True), W.run(d, False)
        finally:
            k._plan_sink = None
        assert labels[0] == "wino_pq256x320" and not labels[1].startswith("wino"), labels
        ref = W.reference(d)
        assert 50.0 < float(ref.std()) < 400.0 and float(ref.abs().max()) > 400.0   # (the operands really are large)
        rel, rel_d = ((y.float() - ref).norm() / ref.norm()).item(), ((direct.float() - ref).norm() / ref.norm()).item()
        assert torch.isfinite(y.float()).all() and torch.isfinite(direct.float()).all()
        assert rel < 3e-3 and rel < 4 * rel_d + 1e-4, (rel, rel_d)


def test_conv3x3_winograd():
    """The Winograd F(2x2, 3x3) route of ca_conv3x3 (ABI v12 w_wino, csrc/ca_conv_wino.h) on the shapes the plan gives it -- 16x16- and
    8x8-latent resnet convolutions, single input and the skip concatenations, with and without the resnet epilogue (bias, time-embedding
    row bias, residual, 1 / output_scale_factor) -- against fp32 torch (animatediff/models/resnet.py:12-20: nn.Conv2d per frame) and
    beside the direct implicit-GEMM form; bit-reproducible; shapes outside its window take the direct form."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import wino_check as W
    k = _k()
    for (images, h, c1, c2, cout) in [(32, 16, 1280, 0, 1280), (32, 16, 1280, 640, 1280), (32, 8, 1280, 1280, 1280), (16, 16, 1280, 640, 640), (32, 16, 640, 0, 1280)]:
        for epi in (True, False):
            d = W.make(images, h, c1, c2, cout, epilogue=epi)
            k._plan_sink = labels = []
            try:
                y = W.run(d, True)
                direct = W.run(d, False)
            finally:
                k._plan_sink = None
            assert labels[0] == "wino_pq256x320" and not labels[1].startswith("wino"), labels
            ref = W.reference(d)
            rel = ((y.float() - ref).norm() / ref.norm()).item()
            rel_d = ((direct.float() - ref).norm() / ref.norm()).item()
            assert torch.isfinite(y.float()).all() and rel < 3e-3, (images, h, c1, c2, cout, epi, rel, rel_d)
            assert torch.equal(y, W.run(d, True))
    # bf16: the transforms run in fp32 arithmetic (no packed bf16 add); the transformed operands carry 8 mantissa bits
    for (images, h, c1, c2, cout) in [(32, 16, 1280, 0, 1280), (32, 8, 1280, 1280, 1280)]:
        d = W.make(images, h, c1, c2, cout, dt=torch.bfloat16, epilogue=True)
        k._plan_sink = labels = []
        try:
            y, direct = W.run(d, True), W.run(d, False)
        finally:
            k._plan_sink = None
        assert labels[0] == "wino_pq256x320" and not labels[1].startswith("wino"), labels
        ref = W.reference(d)
        rel, rel_d = ((y.float() - ref).norm() / ref.norm()).item(), ((direct.float() - ref).norm() / ref.norm()).item()
        assert torch.isfinite(y.float()).all() and rel < 1.2e-2 and rel < 5 * rel_d, (rel, rel_d)
    # Upsample3D: nearest x2 folded into the input transform (resnet.py:67-81), against F.interpolate + conv2d
    import torch.nn.functional as F
    d = W.make(32, 8, 1280, 0, 1280, epilogue=False)
    k._plan_sink = labels = []
    try:
        y = k.conv3x3(d["x"], d["w"], upsample=True, w_wino=d["u"])
        direct = k.conv3x3(d["x"], d["w"], upsample=True)
    finally:
        k._plan_sink = None
    assert labels[0] == "wino_pq256x320" and not labels[1].startswith("wino"), labels
    up = F.interpolate(d["x"].float().permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest")
    ref = F.conv2d(up, d["w"].float().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    assert tuple(y.shape) == (32, 16, 16, 1280)
    assert ((y.float() - ref).norm() / ref.norm()).item() < 3e-3 and ((direct.float() - ref).norm() / ref.norm()).item() < 3e-3
    # outside the window: 640 input channels at 8192 tiles -> the direct kernels, whatever w_wino says
    d = W.make(32, 32, 640, 0, 640, epilogue=False)
    k._plan_sink = labels = []
    try:
        W.run(d, True)
    finally:
        k._plan_sink = None
    assert not labels[0].startswith("wino"), labels


@pytest.mark.gpu
def test_groupnorm_writes_the_winograd_input_transform():
    """ca_groupnorm with wino_v + ca_conv3x3 with x_is_wino_v (ABI v12): GroupNorm + SiLU whose output IS the Winograd route's transformed
    input, so the normalised tensor is never written -- bit-identical to GroupNorm, then the Winograd convolution (the same values go
    through the same packed-fp16 transform), single input and the skip concatenation, 16x16 and 8x8 latents; declined (None) where the
    one-launch GroupNorm or the Winograd route does not apply."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import wino_check as W
    k = _k()
    for (images, h, c1, c2, cout) in [(32, 16, 1280, 0, 1280), (32, 16, 1280, 1280, 1280), (32, 8, 1280, 1280, 1280), (16, 16, 1280, 0, 640)]:
        d = W.make(images, h, c1, c2, cout, epilogue=True)
        g = torch.Generator(device="cpu").manual_seed(5)
        gamma = (1.0 + 0.2 * torch.randn(c1 + c2, generator=g)).to(DEV)
        beta = (0.1 * torch.randn(c1 + c2, generator=g)).to(DEV)
        k._plan_sink = labels = []
        try:
            y = k.group_norm_conv3x3_wino(d["x"], gamma, beta, d["w"], d["u"], x2=d["x2"], act=k.ACT_SILU, eps=1e-6, bias=d["bias"], rowbias=d["rowbias"],
                                          rows_per_group=d["rows_per_group"], residual=d["residual"], post_scale=d["post"])
        finally:
            k._plan_sink = None
        assert y is not None and labels == ["gn_wino_pq256x320"], labels
        hn = k.group_norm(d["x"], gamma, beta, x2=d["x2"], eps=1e-6, act=k.ACT_SILU)
        two = k.conv3x3(hn, d["w"], bias=d["bias"], rowbias=d["rowbias"], rows_per_group=d["rows_per_group"], residual=d["residual"],
                        post_scale=d["post"], w_wino=d["u"])
        assert torch.equal(y, two), f"{int((y != two).sum())} of {y.numel()} elements differ"
        # ... and right: against fp32 torch
        import torch.nn.functional as F
        xin = d["x"].float() if d["x2"] is None else torch.cat([d["x"].float(), d["x2"].float()], dim=-1)
        ref_h = F.silu(F.group_norm(xin.permute(0, 3, 1, 2), 32, gamma, beta, 1e-6)).permute(0, 2, 3, 1)
        dd = dict(d, x=ref_h.to(torch.float16), x2=None)
        dd["w"] = d["w"]
        ref = W.reference(dd)
        assert ((y.float() - ref).norm() / ref.norm()).item() < 3e-3
    # declined: 60 channels per group (C = 1920) is not whole 8-channel chunks; a 32x32 image does not fit the one-launch GroupNorm
    # ... and a 4x4 image is 64 tiles: the convolution keeps the direct (split-K) form, whose workspace must not be mistaken for the route's
    for (images, h, c1, c2, cout) in [(32, 16, 1280, 640, 1280), (8, 32, 1280, 0, 1280), (16, 4, 1280, 0, 1280)]:
        d = W.make(images, h, c1, c2, cout, epilogue=False)
        gamma, beta = torch.ones(c1 + c2, device=DEV), torch.zeros(c1 + c2, device=DEV)
        assert k.group_norm_conv3x3_wino(d["x"], gamma, beta, d["w"], d["u"], x2=d["x2"], act=k.ACT_SILU) is None


@pytest.mark.gpu
@pytest.mark.parametrize("dt,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.2e-2)])
@pytest.mark.parametrize("fr", [8, 32])
def test_tattn_fused_with_output_projection_other_frame_counts(fr, dt, tol):
    """ca_tattn_fused with the output stage for 8 frames (BASELINE config 1: two pixels per MFMA row tile, the cross-pixel quarter of the
    scores masked) and 32 frames (config 5: two row tiles per pixel, 2 x 2 score blocks) -- against fp32 torch (motion_module.py:251-331,
    212-224) and against the unfused path (folded q|k|v GEMM, attention over the frames, output projection)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import tattn_check as T
    from controlanimate_amd import kernels as K
    for (b, tokens, lda, res) in [(2, 4096 * 16 // fr // 2, 320, True), (1, 2048 + 128 // fr * 3, 640, True), (2, 2048, 320, False)]:
        x, w, gamma, beta, pe = T.make(b, tokens, dt, lda=lda, fr=fr)
        wo, bo = T.make_out(dt)
        ref = T.reference(x, w, gamma, beta, pe, b, tokens, fr=fr) @ wo.float().t() + bo[None, :] + (x.float() if res else 0.0)
        K._plan_sink = labels = []
        try:
            y = T.fused_out(x, w, gamma, beta, pe, b, tokens, wo, bo, residual=res, fr=fr)
        finally:
            K._plan_sink = None
        assert y is not None and labels == ["tattn_out128"], (labels, b, tokens)
        assert torch.equal(y, T.fused_out(x, w, gamma, beta, pe, b, tokens, wo, bo, residual=res, fr=fr))
        rel = ((y.float() - ref).norm() / ref.norm()).item()
        assert torch.isfinite(y.float()).all() and rel < tol, (fr, b, tokens, rel)
        # the unfused path on the same operands: folded q|k|v GEMM + attention_temporal + to_out
        xc = x.contiguous()
        wf = (w.float() * gamma[None, :]).to(dt)
        qkv = K.gemm(xc, wf, bias=(w.float() @ beta).contiguous(), ln=(K.RowStats(xc, 1e-5), wf.float().sum(1).contiguous()),
                     rowbias=(pe[:fr] @ w.float().t()).repeat(b, 1).contiguous(), rows_per_group=tokens)
        old = K.gemm(K.attention_temporal(qkv, b, fr, tokens, 8), wo, bias=bo, residual=xc if res else None)
        rel_old = ((old.float() - ref).norm() / ref.norm()).item()
        assert rel < 1.5 * rel_old + 1e-4, (rel, rel_old)
    # without the output stage only 16 frames are taken (the four-wave kernel)
    x, w, gamma, beta, pe = T.make(2, 2048, dt, fr=fr)
    from controlanimate_amd.layers import frag_order_tattn
    assert K.tattn_fused(x, frag_order_tattn(w.float()).to(dt), gamma.contiguous(), (pe + beta[None, :]).contiguous(), 2, fr, 2048, 8, 1e-5, 40 ** -0.5) is None
