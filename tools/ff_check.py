"""Fused feed-forward (ca_ff_fused, csrc/ca_ff_fused.h) against the two ca_gemm calls it replaces and against fp32 torch;
determinism; timing of both inside a hipGraph.
    python tools/ff_check.py [--time-only]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from controlanimate_amd import kernels as K
from controlanimate_amd.layers import frag_order, frag_order2, geglu_interleave

dev = "cuda"


def make(m, dt, seed=3, lda=320):
    g = torch.Generator(device="cpu").manual_seed(seed)
    rn = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev)
    xw = rn(m, lda).to(dt)
    x = xw[:, :320] if lda != 320 else xw
    w1 = geglu_interleave(rn(2560, 320, scale=320 ** -0.5)).to(dt).contiguous()   # (rows value / gate interleaved, as LnFold packs them)
    b1 = rn(2560)
    w2 = rn(320, 1280, scale=1280 ** -0.5).to(dt)
    b2 = rn(320)
    cs = w1.float().sum(1).contiguous()
    w1f = frag_order(w1.float(), True).to(dt).contiguous()
    w2f = frag_order2(w2.float()).to(dt).contiguous()
    return dict(x=x, w1=w1, b1=b1, cs=cs, w2=w2, b2=b2, w1f=w1f, w2f=w2f)


def two_gemms(d, residual=True, frag=True):
    w1 = d["w1"]
    if hasattr(w1, "_frag"):
        del w1._frag
    if frag:
        w1._frag = (d["w1f"], True)
    h = K.gemm(d["x"], w1, bias=d["b1"], geglu=True, ln=(K.RowStats(d["x"], 1e-5), d["cs"]))
    return K.gemm(h, d["w2"], bias=d["b2"], residual=d["x"] if residual else None)


def fused(d, residual=True):
    return K.ff_fused(d["x"], d["w1f"], d["b1"], d["cs"], d["w2f"], d["b2"], 1e-5, residual=d["x"] if residual else None)


def reference(d, residual=True):
    x = d["x"].float()
    dt = d["x"].dtype
    xn = (x - x.mean(1, keepdim=True)) * (x.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
    p = (xn @ d["w1"].float().t() + d["b1"]).to(dt).float()
    h = (p[:, 0::2] * F.gelu(p[:, 1::2])).to(dt).float()
    y = (h @ d["w2"].float().t() + d["b2"]).to(dt).float()
    return y + x if residual else y


def make_out(dt, seed=21):
    """The transformer's proj_out behind the feed-forward: weight [320, 320], bias [320], and the transformer's input as residual."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    wo = (torch.randn(320, 320, generator=g) * 320 ** -0.5 * 1.5).to("cuda").to(dt)
    bo = (torch.randn(320, generator=g) * 0.1).to("cuda")
    return wo, bo


def fused_out(d, wo, bo, res_out, residual=True, wofrag=None):
    """ABI v12: feed-forward + proj_out + bias + residual_out in one launch."""
    from controlanimate_amd.layers import frag_order_wout
    if wofrag is None:
        wofrag = frag_order_wout(wo.float()).to(wo.dtype)
    return K.ff_fused(d["x"], d["w1f"], d["b1"], d["cs"], d["w2f"], d["b2"], 1e-5, residual=d["x"] if residual else None,
                      w_out_frag=wofrag, bias_out=bo, residual_out=res_out)


def two_launch_out(d, wo, bo, res_out, residual=True):
    return K.gemm(fused(d, residual), wo, bias=bo, residual=res_out)


def check():
    bad = 0
    for dt in (torch.float16, torch.bfloat16):
        for (m, lda, res) in ((16384, 320, True), (16384 + 72, 320, True), (131072, 320, True), (20480, 640, False)):
            d = make(m, dt, lda=lda)
            outs = [fused(d, res) for _ in range(3)]
            if outs[0] is None:
                print(f"{dt} m={m}: ca_ff_fused_supported says no   <<<<<< FAIL")
                bad += 1
                continue
            ref, two = reference(d, res), two_gemms(d, res)
            rel = ((outs[0].float() - ref).norm() / ref.norm()).item()
            rel2 = ((two.float() - ref).norm() / ref.norm()).item()
            diff = ((outs[0].float() - two.float()).norm() / ref.norm()).item()
            same = all(torch.equal(outs[0], o) for o in outs[1:])
            tol = 2e-3 if dt == torch.float16 else 1.2e-2
            ok = rel < tol and same and bool(torch.isfinite(outs[0].float()).all()) and diff < tol / 2  # (one operand rounding apart: LN(x) rounded to the activation type in the fused kernel)
            bad += not ok
            print(f"{str(dt)[6:]:9s} m={m:6d} lda={lda} residual={res}: fused rel {rel:.2e}, two GEMMs rel {rel2:.2e}, fused vs two {diff:.2e}, deterministic={same}{'' if ok else '   <<<<<< FAIL'}", flush=True)
    return bad


def timing():
    d = make(131072, torch.float16)
    wo, bo = make_out(torch.float16)
    from controlanimate_amd.layers import frag_order_wout
    wol = frag_order_wout(wo.float()).to(wo.dtype)
    res_out = torch.randn(131072, 320, device="cuda").half()
    for name, fn in (("fused", lambda: fused(d)), ("two GEMMs (ar + pq)", lambda: two_gemms(d)), ("fused", lambda: fused(d)), ("two GEMMs (ar + pq)", lambda: two_gemms(d)),
                     ("fused + proj_out", lambda: fused_out(d, wo, bo, res_out, wofrag=wol)), ("fused, proj_out", lambda: two_launch_out(d, wo, bo, res_out)),
                     ("fused + proj_out", lambda: fused_out(d, wo, bo, res_out, wofrag=wol)), ("fused, proj_out", lambda: two_launch_out(d, wo, bo, res_out))):
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
        us = s.elapsed_time(e) / 30 * 1e3
        print(f"time 131072 rows: {name:22s} {us:7.1f} us  ({2 * 131072 * 320 * 3840 / us * 1e-6:6.1f} TF)", flush=True)


if __name__ == "__main__":
    rc = 0
    if "--time-only" not in sys.argv:
        rc = check()
    timing()
    sys.exit(1 if rc else 0)
