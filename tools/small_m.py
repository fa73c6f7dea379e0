"""The latency-bound GEMMs of the 8x8-latent level (M = 2048) and their neighbours: time per launch and dispatch label.
Experiments build: CA_GEMM_NBUF=3|4 (ring depth of k_gemm_dma), CA_GEMM_BN=64|128 (tile width).
    [CA_HIP_LIB=.../libcontrolanimate_hip_exp.so CA_GEMM_NBUF=3] python tools/small_m.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlanimate_amd import kernels as K

dev, dt = "cuda", torch.float16


def timeit(fn, it=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


rows = []
for (m, n, k, extra) in [(2048, 1280, 1280, "res"), (2048, 1280, 1280, ""), (2048, 3840, 1280, "ln"), (2048, 1280, 5120, "res"), (2048, 10240, 1280, "geglu+ln"),
                         (8192, 1280, 1280, "res"), (8192, 3840, 1280, "ln"), (32768, 640, 640, "res"), (32, 1280, 320, "")]:
    a = torch.randn(m, k, device=dev).to(dt)
    w = (torch.randn(n, k, device=dev) * k ** -0.5).to(dt)
    kw = dict(bias=torch.randn(n, device=dev))
    if "res" in extra:
        kw["residual"] = torch.randn(m, n, device=dev).to(dt)
    if "geglu" in extra:
        kw["geglu"] = True
    if "ln" in extra:
        st = torch.stack([a.float().mean(1), (a.float().var(1, unbiased=False) + 1e-5).rsqrt()], 1).contiguous()
        kw["ln"] = (st, w.float().sum(1).contiguous())
    K._plan_sink = []
    ref = K.gemm(a, w, **kw)
    lab = K._plan_sink[-1]
    K._plan_sink = None
    # correctness against fp32
    x = a.float()
    if "ln" in extra:
        x = (x - kw["ln"][0][:, :1]) * kw["ln"][0][:, 1:]
    y = x @ w.float().t() + kw["bias"]
    y = y.to(dt).float()
    if "res" in extra:
        y = y + kw["residual"].float()
    if "geglu" in extra:
        y = y[:, 0::2] * torch.nn.functional.gelu(y[:, 1::2])
    rel = ((ref.float() - y).norm() / y.norm()).item()
    us = timeit(lambda: K.gemm(a, w, **kw))
    print(f"{m}x{n}x{k} {extra:9s} {lab:18s} {us:7.1f} us  {2 * m * n * k / us * 1e-6:6.1f} TF  rel {rel:.1e}", flush=True)
