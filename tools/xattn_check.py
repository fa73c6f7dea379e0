"""Fused text cross-attention (ca_xattn_fused.h): correctness against fp32 torch and against the two-launch path (folded q GEMM +
attention over the 77 text tokens), the library's packing kernels against Python, determinism and timing (inside a hipGraph).
    python tools/xattn_check.py [--time-only]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from controlanimate_amd import kernels as K
from controlanimate_amd.layers import frag_order_xattn

dev = "cuda"
HEADS, D, CH = 8, 40, 320


def make(images, tokens, L, kvb, dt, seed=5, lda=CH):
    g = torch.Generator(device="cpu").manual_seed(seed)
    rn = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev)
    xw = (rn(images * tokens, lda) * 1.5 + 0.3).to(dt)
    x = xw[:, :CH]
    wq = rn(CH, CH, scale=CH ** -0.5 * 2.0)
    gamma, beta = 1.0 + 0.2 * rn(CH), 0.1 * rn(CH)
    kv = rn(kvb * L, 2 * CH).to(dt)
    return x, wq, gamma, beta, kv


def reference(x, wq, gamma, beta, kv, images, tokens, L, nk, fpk, kvb, dt):
    wf = (wq * gamma[None, :]).to(dt).float()  # the folded weight as the kernels hold it
    bias = wq @ beta
    n = F.layer_norm(x.float(), (CH,), None, None, 1e-5)
    q = (n @ wf.t() + bias).view(images, tokens, HEADS, D).transpose(1, 2)
    idx = (torch.arange(images, device=dev) // fpk) % kvb
    k = kv[:, :CH].float().view(kvb, L, HEADS, D)[:, :nk].transpose(1, 2)[idx]
    v = kv[:, CH:].float().view(kvb, L, HEADS, D)[:, :nk].transpose(1, 2)[idx]
    o = F.softmax(q @ k.transpose(-1, -2) * D ** -0.5, dim=-1) @ v
    return o.transpose(1, 2).reshape(images * tokens, CH)


def operands(x, wq, gamma, beta, dt):
    wf = (wq * gamma[None, :]).to(dt)
    return wf, wf.float().sum(1).contiguous(), (wq @ beta).contiguous()


def two_launch(x, wq, gamma, beta, kv, images, tokens, L, nk, fpk, kvb):
    wf, cs, bias = operands(x, wq, gamma, beta, x.dtype)
    xc = x.contiguous()
    q = K.gemm(xc, wf, bias=bias, ln=(K.RowStats(xc, 1e-5), cs))
    return K.attention_cross(q, kv, images, tokens, HEADS, nk, L, fpk, kv_mod=kvb)


def fused(x, wq, gamma, beta, kv, images, tokens, L, nk, fpk, kvb, packed=None):
    wf, cs, bias = operands(x, wq, gamma, beta, x.dtype)
    if packed is None:
        packed = (frag_order_xattn(wf.float()).to(x.dtype), K.xattn_pack_kv(kv, kvb, L, nk, D ** -0.5))
    return K.xattn_fused(x, packed[0], bias, packed[1], images, tokens, fpk, kvb, nk, 1e-5)


def make_out(dt, seed=11):
    g = torch.Generator(device="cpu").manual_seed(seed)
    wo = (torch.randn(CH, CH, generator=g) * CH ** -0.5 * 1.5).to(dev).to(dt)
    bo = (torch.randn(CH, generator=g) * 0.1).to(dev)
    return wo, bo


def fused_out(x, wq, gamma, beta, kv, images, tokens, L, nk, fpk, kvb, wo, bo, packed=None, wofrag=None, residual=True):
    """ABI v12: attention + output projection + bias + residual (the block's own input) in one launch."""
    from controlanimate_amd.layers import frag_order_wout
    wf, cs, bias = operands(x, wq, gamma, beta, x.dtype)
    if packed is None:
        packed = (frag_order_xattn(wf.float()).to(x.dtype), K.xattn_pack_kv(kv, kvb, L, nk, D ** -0.5))
    if wofrag is None:
        wofrag = frag_order_wout(wo.float()).to(x.dtype)
    return K.xattn_fused(x, packed[0], bias, packed[1], images, tokens, fpk, kvb, nk, 1e-5, w_out_frag=wofrag, bias_out=bo,
                         residual=x if residual else None)


def two_launch_out(x, wq, gamma, beta, kv, images, tokens, L, nk, fpk, kvb, wo, bo, packed=None, residual=True):
    o = fused(x, wq, gamma, beta, kv, images, tokens, L, nk, fpk, kvb, packed)
    return K.gemm(o, wo, bias=bo, residual=x if residual else None)


def reference_ip(x, wq, gamma, beta, kv, kvip, images, tokens, L, nk, nip, fpk, kvb, dt, ip_scale):
    """fp32: o_text + ip_scale * o_ip (modules/attention_processor.py:433-477; the image-prompt tokens are the last nip context rows)."""
    wf = (wq * gamma[None, :]).to(dt).float()
    n = F.layer_norm(x.float(), (CH,), None, None, 1e-5)
    q = (n @ wf.t() + wq @ beta).view(images, tokens, HEADS, D).transpose(1, 2)
    idx = (torch.arange(images, device=dev) // fpk) % kvb

    def att(src, lo, hi):
        k = src[:, :CH].float().view(kvb, L, HEADS, D)[:, lo:hi].transpose(1, 2)[idx]
        v = src[:, CH:].float().view(kvb, L, HEADS, D)[:, lo:hi].transpose(1, 2)[idx]
        return F.softmax(q @ k.transpose(-1, -2) * D ** -0.5, dim=-1) @ v
    o = att(kv, 0, nk) + ip_scale * att(kvip, L - nip, L)
    return o.transpose(1, 2).reshape(images * tokens, CH)


def fused_ip_out(x, wq, gamma, beta, kv, kvip, images, tokens, L, nk, nip, fpk, kvb, wo, bo, ip_scale, residual=True):
    """ABI v13: text attention + the IP-Adapter's image-prompt attention + output projection + bias + residual in one launch."""
    from controlanimate_amd.layers import frag_order_wout
    wf, cs, bias = operands(x, wq, gamma, beta, x.dtype)
    kvf = K.xattn_pack_kv(kv, kvb, L, nk, D ** -0.5)
    kvf_ip = K.xattn_pack_kv(kvip, kvb, L, nip, D ** -0.5, row_offset=L - nip)
    return K.xattn_fused(x, frag_order_xattn(wf.float()).to(x.dtype), bias, kvf, images, tokens, fpk, kvb, nk, 1e-5,
                         w_out_frag=frag_order_wout(wo.float()).to(x.dtype), bias_out=bo, residual=x if residual else None,
                         kv_frag_ip=kvf_ip, nk_ip=nip, ip_scale=ip_scale)


def separate_ip_out(x, wq, gamma, beta, kv, kvip, images, tokens, L, nk, nip, fpk, kvb, wo, bo, ip_scale, residual=True):
    """What the product ran before: folded q GEMM, attention over the text tokens, the accumulating attention over the image-prompt
    tokens (IPAttnProcessor2_0.ip_branch), to_out."""
    wf, cs, bias = operands(x, wq, gamma, beta, x.dtype)
    xc = x.contiguous()
    q = K.gemm(xc, wf, bias=bias, ln=(K.RowStats(xc, 1e-5), cs))
    o = K.attention_cross(q, kv, images, tokens, HEADS, nk, L, fpk, kv_mod=kvb)
    o = K.attention_cross(q, kvip, images, tokens, HEADS, nip, L, fpk, out=o, out_scale=float(ip_scale), accumulate=True, kv_row_offset=L - nip, kv_mod=kvb)
    return K.gemm(o, wo, bias=bo, residual=xc if residual else None)


def check_ip():
    bad = 0
    for dt in (torch.float16, torch.bfloat16):
        for (images, tokens, L, nk, nip, fpk, kvb, lda, res, sc) in [(32, 4096, 81, 77, 4, 16, 2, 320, True, 1.0), (8, 2048, 81, 77, 4, 4, 2, 320, True, 0.6),
                                                                     (6, 3072, 88, 72, 16, 2, 3, 640, True, 1.0), (16, 1024, 78, 77, 1, 8, 2, 320, False, 0.5)]:
            x, wq, gamma, beta, kv = make(images, tokens, L, kvb, dt, lda=lda)
            kvip = make(images, tokens, L, kvb, dt, seed=23, lda=lda)[4]
            wo, bo = make_out(dt)
            ref = (reference_ip(x, wq, gamma, beta, kv, kvip, images, tokens, L, nk, nip, fpk, kvb, dt, sc) @ wo.float().t() + bo[None, :]
                   + (x.float() if res else 0.0))
            outs = [fused_ip_out(x, wq, gamma, beta, kv, kvip, images, tokens, L, nk, nip, fpk, kvb, wo, bo, sc, residual=res) for _ in range(3)]
            if outs[0] is None:
                print(f"ip  {str(dt)[6:]:9s} images={images} tokens={tokens}: not taken by the library   <<<<<< FAIL")
                bad += 1
                continue
            old = separate_ip_out(x, wq, gamma, beta, kv, kvip, images, tokens, L, nk, nip, fpk, kvb, wo, bo, sc, residual=res)
            rel = ((outs[0].float() - ref).norm() / ref.norm()).item()
            rel_old = ((old.float() - ref).norm() / ref.norm()).item()
            det = all(torch.equal(outs[0], o) for o in outs[1:])
            tol = 2e-3 if dt == torch.float16 else 1.2e-2
            ok = rel < tol and rel < 1.5 * rel_old + 1e-4 and det and bool(torch.isfinite(outs[0].float()).all())
            bad += not ok
            print(f"ip  {str(dt)[6:]:9s} images={images} tokens={tokens:5d} L={L} nk={nk} nip={nip} scale={sc} lda={lda} residual={res}: rel {rel:.2e} "
                  f"(separate launches {rel_old:.2e}) deterministic={det}{'' if ok else '   <<<<<< FAIL'}", flush=True)
    return bad


def check_out():
    bad = 0
    for dt in (torch.float16, torch.bfloat16):
        for (images, tokens, L, nk, fpk, kvb, lda, res) in [(32, 4096, 77, 77, 16, 2, 320, True), (8, 2048, 77, 77, 4, 2, 320, True),
                                                            (6, 3072, 81, 77, 2, 3, 640, True), (16, 1024, 70, 70, 8, 2, 320, False)]:
            x, wq, gamma, beta, kv = make(images, tokens, L, kvb, dt, lda=lda)
            wo, bo = make_out(dt)
            ref = reference(x, wq, gamma, beta, kv, images, tokens, L, nk, fpk, kvb, dt) @ wo.float().t() + bo[None, :] + (x.float() if res else 0.0)
            outs = [fused_out(x, wq, gamma, beta, kv, images, tokens, L, nk, fpk, kvb, wo, bo, residual=res) for _ in range(3)]
            if outs[0] is None:
                print(f"out {str(dt)[6:]:9s} images={images} tokens={tokens}: not taken by the library   <<<<<< FAIL")
                bad += 1
                continue
            old = two_launch_out(x, wq, gamma, beta, kv, images, tokens, L, nk, fpk, kvb, wo, bo, residual=res)
            rel = ((outs[0].float() - ref).norm() / ref.norm()).item()
            rel_old = ((old.float() - ref).norm() / ref.norm()).item()
            diff = int((outs[0] != old).sum())
            det = all(torch.equal(outs[0], o) for o in outs[1:])
            tol = 2e-3 if dt == torch.float16 else 1.2e-2
            ok = rel < tol and rel < 1.5 * rel_old + 1e-4 and det and bool(torch.isfinite(outs[0].float()).all())
            bad += not ok
            print(f"out {str(dt)[6:]:9s} images={images} tokens={tokens:5d} L={L} lda={lda} residual={res}: rel {rel:.2e} (two launches {rel_old:.2e}; {diff} of "
                  f"{old.numel()} elements differ from them) deterministic={det}{'' if ok else '   <<<<<< FAIL'}", flush=True)
    return bad


def check():
    bad = 0
    for dt in (torch.float16, torch.bfloat16):
        for (images, tokens, L, nk, fpk, kvb, lda) in [(32, 4096, 77, 77, 16, 2, 320), (8, 2048, 77, 77, 4, 2, 320), (6, 3072, 81, 77, 2, 3, 640), (32, 6144, 77, 77, 16, 2, 320),
                                                        (16, 1024, 80, 80, 16, 1, 320), (16, 1024, 70, 70, 8, 2, 320)]:
            x, wq, gamma, beta, kv = make(images, tokens, L, kvb, dt, lda=lda)
            ref = reference(x, wq, gamma, beta, kv, images, tokens, L, nk, fpk, kvb, dt)
            wf, cs, bias = operands(x, wq, gamma, beta, dt)
            wl = torch.empty(122880, device=dev, dtype=dt)
            K.check(K.lib().ca_xattn_pack_w(wf.data_ptr(), 320, 320, wl.data_ptr(), K._stream()), "ca_xattn_pack_w")
            same_pack = torch.equal(wl, frag_order_xattn(wf.float()).to(dt))
            kvf = K.xattn_pack_kv(kv, kvb, L, nk, D ** -0.5)
            outs = [fused(x, wq, gamma, beta, kv, images, tokens, L, nk, fpk, kvb, (wl, kvf)) for _ in range(3)]
            if outs[0] is None:
                print(f"{str(dt)[6:]:9s} images={images} tokens={tokens}: not taken by the library   <<<<<< FAIL")
                bad += 1
                continue
            old = two_launch(x, wq, gamma, beta, kv, images, tokens, L, nk, fpk, kvb)
            rel = ((outs[0].float() - ref).norm() / ref.norm()).item()
            rel_old = ((old.float() - ref).norm() / ref.norm()).item()
            det = all(torch.equal(outs[0], o) for o in outs[1:])
            tol = 2e-3 if dt == torch.float16 else 1.2e-2
            ok = rel < tol and det and same_pack and bool(torch.isfinite(outs[0].float()).all())
            bad += not ok
            print(f"{str(dt)[6:]:9s} images={images:2d} tokens={tokens:5d} L={L} nk={nk} fpk={fpk} kvb={kvb} lda={lda}: rel {rel:.2e} (two launches {rel_old:.2e}) "
                  f"deterministic={det} pack={same_pack}{'' if ok else '   <<<<<< FAIL'}", flush=True)
    return bad


def timeit(fn):
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
        for (images, tokens) in [(32, 4096), (32, 6144)]:
            L, nk, fpk, kvb = 77, 77, 16, 2
            x, wq, gamma, beta, kv = make(images, tokens, L, kvb, dt)
            wf, cs, bias = operands(x, wq, gamma, beta, dt)
            packed = (frag_order_xattn(wf.float()).to(dt), K.xattn_pack_kv(kv, kvb, L, nk, D ** -0.5))

            def old():
                q = K.gemm(x, wf, bias=bias, ln=(K.RowStats(x, 1e-5), cs))
                return K.attention_cross(q, kv, images, tokens, HEADS, nk, L, fpk, kv_mod=kvb)
            row = []
            from controlanimate_amd.layers import frag_order_wout
            wo, bo = make_out(dt)
            wol = frag_order_wout(wo.float()).to(dt)
            for _ in range(2):
                row.append(("fused", timeit(lambda: K.xattn_fused(x, packed[0], bias, packed[1], images, tokens, fpk, kvb, nk, 1e-5))))
                row.append(("gemm+attn", timeit(old)))
                row.append(("fused+out", timeit(lambda: K.xattn_fused(x, packed[0], bias, packed[1], images, tokens, fpk, kvb, nk, 1e-5, w_out_frag=wol, bias_out=bo, residual=x))))
                row.append(("fused, to_out", timeit(lambda: K.gemm(K.xattn_fused(x, packed[0], bias, packed[1], images, tokens, fpk, kvb, nk, 1e-5), wo, bias=bo, residual=x))))
            print(f"time {str(dt)[6:]:9s} rows {images * tokens:7d}: " + "  ".join(f"{n} {us:7.1f} us" for n, us in row), flush=True)
            if tokens == 4096:  # the IP-Adapter's site (config 4): 77 text + 4 image-prompt tokens
                L2_, nip = 81, 4
                kv2 = make(images, tokens, L2_, kvb, dt)[4]
                kvip = make(images, tokens, L2_, kvb, dt, seed=23)[4]
                kvf, kvf_ip = K.xattn_pack_kv(kv2, kvb, L2_, 77, D ** -0.5), K.xattn_pack_kv(kvip, kvb, L2_, nip, D ** -0.5, row_offset=L2_ - nip)

                def sep():
                    q = K.gemm(x, wf, bias=bias, ln=(K.RowStats(x, 1e-5), cs))
                    o = K.attention_cross(q, kv2, images, tokens, HEADS, 77, L2_, fpk, kv_mod=kvb)
                    o = K.attention_cross(q, kvip, images, tokens, HEADS, nip, L2_, fpk, out=o, out_scale=1.0, accumulate=True, kv_row_offset=L2_ - nip, kv_mod=kvb)
                    return K.gemm(o, wo, bias=bo, residual=x)
                row = []
                for _ in range(2):
                    row.append(("fused+ip+out", timeit(lambda: K.xattn_fused(x, packed[0], bias, kvf, images, tokens, fpk, kvb, 77, 1e-5, w_out_frag=wol, bias_out=bo,
                                                                             residual=x, kv_frag_ip=kvf_ip, nk_ip=nip, ip_scale=1.0))))
                    row.append(("gemm, attn, attn ip, to_out", timeit(sep)))
                print(f"time {str(dt)[6:]:9s} rows {images * tokens:7d} (IP tokens): " + "  ".join(f"{n} {us:7.1f} us" for n, us in row), flush=True)


if __name__ == "__main__":
    rc = 0
    if "--time-only" not in sys.argv:
        rc = check() + check_out() + check_ip()
    timing()
    sys.exit(1 if rc else 0)
