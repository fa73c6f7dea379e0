"""Micro-benchmark of GroupNorm / LayerNorm at UNet shapes.  python tools/bench_norm.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlanimate_amd import kernels as K
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
for (img, h, c) in [(32, 64, 320), (32, 32, 640), (32, 16, 1280), (32, 8, 1280), (32, 64, 960), (32, 32, 1920)]:
    x = torch.randn(img, h, h, c, device="cuda").half(); g = torch.ones(c, device="cuda"); b = torch.zeros(c, device="cuda")
    ms = timeit(lambda: K.group_norm(x, g, b, act=1))
    print(f"groupnorm {img}x{h}x{h}x{c}: {ms*1e3:7.1f} us  {3*x.numel()*2/ms/1e9:6.2f} TB/s (2 reads + 1 write)")
for (rows, c) in [(131072, 320), (32768, 640), (8192, 1280)]:
    x = torch.randn(rows, c, device="cuda").half(); g = torch.ones(c, device="cuda"); b = torch.zeros(c, device="cuda")
    ms = timeit(lambda: K.layer_norm(x, g, b))
    print(f"layernorm {rows}x{c}: {ms*1e3:7.1f} us  {2*x.numel()*2/ms/1e9:6.2f} TB/s")
