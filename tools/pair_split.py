"""GEGLU -> FF-out at the 64x64-latent level (131072 rows) in one piece or in row slices: does the 335 MB intermediate come
back out of the memory-side cache (256 MB) when a slice of it fits?
    python tools/pair_split.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlanimate_amd import kernels as K

dev, dt, m = "cuda", torch.float16, 131072
x = torch.randn(m, 320, device=dev).to(dt)
w1 = (torch.randn(2560, 320, device=dev) * 320 ** -0.5).to(dt)
b1 = torch.randn(2560, device=dev)
cs = w1.float().sum(1).contiguous()
K.attach_w_frag(w1, True)
w2 = (torch.randn(320, 1280, device=dev) * 1280 ** -0.5).to(dt)
b2 = torch.randn(320, device=dev)
h = torch.empty(m, 1280, device=dev, dtype=dt)
y = torch.empty(m, 320, device=dev, dtype=dt)


def run(parts):
    step = m // parts
    for i in range(parts):
        xs, hs, ys = x[i * step:(i + 1) * step], h[i * step:(i + 1) * step], y[i * step:(i + 1) * step]
        K.gemm(xs, w1, bias=b1, geglu=True, ln=(K.RowStats(xs, 1e-5), cs), out=hs)
        K.gemm(hs, w2, bias=b2, residual=xs, out=ys)


for parts in (1, 2, 4, 8, 1, 2, 4):
    for _ in range(3):
        run(parts)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            run(parts)
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    print(f"{parts} slice(s): GEGLU + FF-out {s.elapsed_time(e) / 30 * 1e3:.1f} us per pair", flush=True)
