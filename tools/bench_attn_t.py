"""Temporal attention micro-benchmark (16 frames per pixel): python tools/bench_attn_t.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlanimate_amd import kernels as K
from tools.bench_gemm import timeit
for (b, f, n, heads, d) in [(2, 16, 4096, 8, 40), (2, 16, 1024, 8, 80), (2, 16, 256, 8, 160), (2, 16, 64, 8, 160)]:
    c = heads * d
    qkv = torch.randn(b * f * n, 3 * c, device="cuda").half()
    ms = timeit(lambda: K.attention_temporal(qkv, b, f, n, heads))
    byts = qkv.numel() * 2 + b * f * n * c * 2
    print(f"temporal b={b} f={f} n={n} heads={heads} d={d}: {ms*1e3:7.1f} us  {byts/ms/1e9:.2f} TB/s  [T16={os.environ.get('CA_ATTN_T16','1')}]")
