"""Micro-benchmark of the text cross-attention shapes (77 keys) under CA_ATTN_VAR: python tools/bench_attn_cross.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlanimate_amd import kernels as K
for (images, heads, d, n, L) in [(32, 8, 40, 4096, 77), (32, 8, 80, 1024, 77), (32, 8, 160, 256, 77), (32, 8, 40, 4096, 81), (32, 8, 40, 6144, 77), (32, 8, 80, 1000, 77), (16, 8, 80, 1536, 70)]:
    c = heads * d
    q = torch.randn(images * n, c, device="cuda").half()
    kv = torch.randn(2 * L, 2 * c, device="cuda").half()
    f = lambda: K.attention_cross(q, kv, images, n, heads, L, L, 16, kv_mod=2)
    for _ in range(3): o = f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): o = f()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    ref = None
    if n <= 4096:
        qh = q.float().view(images, n, heads, d).transpose(1, 2)
        kk = kv[:, :c].float().view(2, L, heads, d).transpose(1, 2)[torch.arange(images) // 16 % 2]
        vv = kv[:, c:].float().view(2, L, heads, d).transpose(1, 2)[torch.arange(images) // 16 % 2]
        ref = torch.softmax(qh @ kk.transpose(-1, -2) * d ** -0.5, -1) @ vv
        ref = ref.transpose(1, 2).reshape(images * n, c)
        err = ((o.float() - ref).norm() / ref.norm()).item()
    else:
        err = float("nan")
    print(f"cross images={images} d={d} N={n} L={L}: {us:7.1f} us  ({2*q.numel()*2/us/1e6:.2f} TB/s of Q+O)  rel {err:.1e}  var={os.environ.get('CA_ATTN_VAR','0')} short={os.environ.get('CA_ATTN_SHORT','1')}")
