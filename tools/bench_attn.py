"""Micro-benchmark of ca_attention at the UNet's hottest shape (spatial self-attention, 64x64 latent).
   python tools/bench_attn.py [images heads head_dim tokens]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlanimate_amd import kernels as K
images, heads, d, n = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (32, 8, 40, 4096)))
c = heads * d
qkv = (torch.randn(images * n, 3 * c, device="cuda") * 1.0).half()
for _ in range(3):
    o = K.attention_spatial(qkv, images, n, heads)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
it = 10
for _ in range(it):
    o = K.attention_spatial(qkv, images, n, heads)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / it
fl = 4.0 * n * n * d * heads * images
print(f"attn images={images} heads={heads} d={d} N={n}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TFLOP/s (algorithmic)  var={os.environ.get('CA_ATTN_VAR','0')}")
