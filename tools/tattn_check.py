"""Fused temporal attention (ca_tattn_fused.h): correctness against fp32 torch and against the two-launch path (folded q|k|v GEMM +
attention over the frames), the library's weight packing against layers.frag_order_tattn, determinism and timing -- one process.
    python tools/tattn_check.py [--time-only]
"""
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from controlanimate_amd import kernels as K
from controlanimate_amd.layers import frag_order_tattn

dev = "cuda"
HEADS, D, CH, FR = 8, 40, 320, 16


def make(b, tokens, dt, seed=3, lda=CH, fr=FR):
    g = torch.Generator(device="cpu").manual_seed(seed)
    rn = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev)
    rows = b * fr * tokens
    xw = (rn(rows, lda) * 1.5 + 0.3).to(dt)
    x = xw[:, :CH]
    w = rn(3 * CH, CH, scale=CH ** -0.5 * 2.0).to(dt)
    gamma, beta = 1.0 + 0.2 * rn(CH), 0.1 * rn(CH)
    pos = torch.arange(32).unsqueeze(1)
    div = torch.exp(torch.arange(0, CH, 2) * (-math.log(10000.0) / CH))
    pe = torch.zeros(32, CH)
    pe[:, 0::2], pe[:, 1::2] = torch.sin(pos * div), torch.cos(pos * div)
    return x, w, gamma, beta, pe.to(dev)


def reference(x, w, gamma, beta, pe, b, tokens, fr=FR):
    """fp32 throughout (rounding only where the inputs already are rounded)."""
    xf = x.float()
    n = F.layer_norm(xf, (CH,), gamma, beta, 1e-5)
    n = n.view(b, fr, tokens, CH) + pe[:fr].view(1, fr, 1, CH)
    qkv = n @ w.float().t()                                  # [b, f, n, 960]
    q, k, v = (t.reshape(b, fr, tokens, HEADS, D).permute(0, 2, 3, 1, 4) for t in qkv.split(CH, dim=-1))  # [b, n, h, f, d]
    o = F.softmax(q @ k.transpose(-1, -2) * D ** -0.5, dim=-1) @ v
    return o.permute(0, 3, 1, 2, 4).reshape(b * fr * tokens, CH)


def two_launch(x, w, gamma, beta, pe, b, tokens):
    """The path the product ran before: LayerNorm folded into the q|k|v GEMM (+ per-frame row bias), then attention over the frames."""
    dt = x.dtype
    x = x.contiguous()
    wf = (w.float() * gamma[None, :]).to(dt)
    cs = wf.float().sum(1).contiguous()
    bias = (w.float() @ beta).contiguous()
    rb = (pe[:FR] @ w.float().t()).repeat(b, 1).contiguous()
    K.attach_w_frag(wf, False)
    qkv = K.gemm(x, wf, bias=bias, ln=(K.RowStats(x, 1e-5), cs), rowbias=rb, rows_per_group=tokens)
    return K.attention_temporal(qkv, b, FR, tokens, HEADS)


def fused(x, w, gamma, beta, pe, b, tokens, wfrag=None):
    if wfrag is None:
        wfrag = frag_order_tattn(w.float()).to(x.dtype)
    bp = (pe + beta[None, :]).contiguous()
    return K.tattn_fused(x, wfrag, gamma.contiguous(), bp, b, FR, tokens, HEADS, 1e-5, D ** -0.5)


def make_out(dt, seed=11):
    """to_out[0] of the attention: weight [320, 320], bias [320]."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    wo = (torch.randn(CH, CH, generator=g) * CH ** -0.5 * 1.5).to(dev).to(dt)
    bo = (torch.randn(CH, generator=g) * 0.1).to(dev)
    return wo, bo


def fused_out(x, w, gamma, beta, pe, b, tokens, wo, bo, wfrag=None, wofrag=None, residual=True, fr=FR):
    """ABI v12: attention + output projection + bias + residual (the block's own input) in one launch."""
    from controlanimate_amd.layers import frag_order_wout
    if wfrag is None:
        wfrag = frag_order_tattn(w.float()).to(x.dtype)
    if wofrag is None:
        wofrag = frag_order_wout(wo.float()).to(x.dtype)
    bp = (pe + beta[None, :]).contiguous()
    return K.tattn_fused(x, wfrag, gamma.contiguous(), bp, b, fr, tokens, HEADS, 1e-5, D ** -0.5, w_out_frag=wofrag, bias_out=bo,
                         residual=x if residual else None)


def two_launch_out(x, w, gamma, beta, pe, b, tokens, wo, bo, wfrag=None, residual=True):
    """What the fused output stage replaces: ca_tattn_fused, then the to_out GEMM (+ bias + residual)."""
    o = fused(x, w, gamma, beta, pe, b, tokens, wfrag)
    return K.gemm(o, wo, bias=bo, residual=x if residual else None)


def check_out():
    from controlanimate_amd.layers import frag_order_wout
    bad = 0
    for dt in (torch.float16, torch.bfloat16):
        for (b, tokens, lda, res) in [(2, 4096, 320, True), (1, 1024, 320, True), (3, 1032, 640, True), (2, 2048, 320, False)]:
            x, w, gamma, beta, pe = make(b, tokens, dt, lda=lda)
            wo, bo = make_out(dt)
            ref = reference(x, w, gamma, beta, pe, b, tokens) @ wo.float().t() + bo[None, :] + (x.float() if res else 0.0)
            wol = torch.empty(102400, device=dev, dtype=dt)
            K.check(K.lib().ca_pack_w_out(wo.data_ptr(), 320, 320, wol.data_ptr(), K._stream()), "ca_pack_w_out")
            same_pack = torch.equal(wol, frag_order_wout(wo.float()).to(dt))
            outs = [fused_out(x, w, gamma, beta, pe, b, tokens, wo, bo, wofrag=wol, residual=res) for _ in range(3)]
            if outs[0] is None:
                print(f"out {str(dt)[6:]:9s} b={b} tokens={tokens}: not taken by the library   <<<<<< FAIL")
                bad += 1
                continue
            old = two_launch_out(x, w, gamma, beta, pe, b, tokens, wo, bo, residual=res)
            rel = ((outs[0].float() - ref).norm() / ref.norm()).item()
            rel_old = ((old.float() - ref).norm() / ref.norm()).item()
            diff = int((outs[0] != old).sum())
            det = all(torch.equal(outs[0], o) for o in outs[1:])
            tol = 2e-3 if dt == torch.float16 else 1.2e-2
            ok = rel < tol and rel < 1.5 * rel_old + 1e-4 and det and same_pack and bool(torch.isfinite(outs[0].float()).all())
            bad += not ok
            print(f"out {str(dt)[6:]:9s} b={b} tokens={tokens:5d} lda={lda} residual={res}: rel {rel:.2e} (two launches {rel_old:.2e}; {diff} of {old.numel()} elements "
                  f"differ from them) deterministic={det} pack={same_pack}{'' if ok else '   <<<<<< FAIL'}", flush=True)
    return bad


def check():
    bad = 0
    for dt in (torch.float16, torch.bfloat16):
        for (b, tokens, lda) in [(2, 4096, 320), (1, 1024, 320), (3, 1032, 640), (2, 6144, 320)]:
            x, w, gamma, beta, pe = make(b, tokens, dt, lda=lda)
            ref = reference(x, w, gamma, beta, pe, b, tokens)
            # the library's packing kernel == the Python packing
            wl = torch.empty(368640, device=dev, dtype=dt)
            K.check(K.lib().ca_pack_w_tattn(w.data_ptr(), 960, 320, wl.data_ptr(), K._stream()), "ca_pack_w_tattn")
            same_pack = torch.equal(wl, frag_order_tattn(w.float()).to(dt))
            outs = [fused(x, w, gamma, beta, pe, b, tokens, wl) for _ in range(3)]
            if outs[0] is None:
                print(f"{str(dt)[6:]:9s} b={b} tokens={tokens}: not taken by the library   <<<<<< FAIL")
                bad += 1
                continue
            old = two_launch(x, w, gamma, beta, pe, b, tokens)
            rel = ((outs[0].float() - ref).norm() / ref.norm()).item()
            rel_old = ((old.float() - ref).norm() / ref.norm()).item()
            mx = (outs[0].float() - ref).abs().max().item()
            det = all(torch.equal(outs[0], o) for o in outs[1:])
            tol = 2e-3 if dt == torch.float16 else 1.2e-2
            ok = rel < tol and det and same_pack and bool(torch.isfinite(outs[0].float()).all())
            bad += not ok
            print(f"{str(dt)[6:]:9s} b={b} tokens={tokens:5d} lda={lda}: rel {rel:.2e} (two launches {rel_old:.2e}) max abs {mx:.2e} deterministic={det} "
                  f"pack={same_pack}{'' if ok else '   <<<<<< FAIL'}", flush=True)
    return bad


def timeit(fn):
    """us per call inside a hipGraph (no host launch gaps)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / 30 * 1e3


def timing():
    for dt in (torch.float16, torch.bfloat16):
        for (b, tokens) in [(2, 4096), (2, 6144)]:
            x, w, gamma, beta, pe = make(b, tokens, dt)
            wl = frag_order_tattn(w.float()).to(dt)
            bp = (pe + beta[None, :]).contiguous()
            wf = (w.float() * gamma[None, :]).to(dt)
            cs = wf.float().sum(1).contiguous()
            bias = (w.float() @ beta).contiguous()
            rb = (pe[:FR] @ w.float().t()).repeat(b, 1).contiguous()
            K.attach_w_frag(wf, False)

            def old():
                qkv = K.gemm(x, wf, bias=bias, ln=(K.RowStats(x, 1e-5), cs), rowbias=rb, rows_per_group=tokens)
                return K.attention_temporal(qkv, b, FR, tokens, HEADS)
            row = []
            wo, bo = make_out(dt)
            from controlanimate_amd.layers import frag_order_wout
            wol = frag_order_wout(wo.float()).to(dt)
            for _ in range(2):
                row.append(("fused", timeit(lambda: K.tattn_fused(x, wl, gamma, bp, b, FR, tokens, HEADS, 1e-5, D ** -0.5))))
                row.append(("gemm+attn", timeit(old)))
                row.append(("fused+out", timeit(lambda: K.tattn_fused(x, wl, gamma, bp, b, FR, tokens, HEADS, 1e-5, D ** -0.5, w_out_frag=wol, bias_out=bo, residual=x))))
                row.append(("fused, to_out", timeit(lambda: K.gemm(K.tattn_fused(x, wl, gamma, bp, b, FR, tokens, HEADS, 1e-5, D ** -0.5), wo, bias=bo, residual=x))))
            print(f"time {str(dt)[6:]:9s} rows {b * FR * tokens:7d}: " + "  ".join(f"{n} {us:7.1f} us" for n, us in row), flush=True)


if __name__ == "__main__":
    rc = 0
    if "--time-only" not in sys.argv:
        rc = check() + check_out()
    timing()
    sys.exit(1 if rc else 0)
