"""Producer -> consumer pairs through the memory-side cache: does the order in which the producer writes its rows (first to last
or last to first, CA_PP_DBG=7 in an experiments build: k_gemm_ar only) change what the consumer's reads cost?
    CA_HIP_LIB=.../libcontrolanimate_hip_exp.so [CA_PP_DBG=7] python tools/pair_order.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlanimate_amd import kernels as K

dev = "cuda"
dt = torch.float16
m = 131072
x = torch.randn(m, 320, device=dev).to(dt)
w1 = (torch.randn(2560, 320, device=dev) * 320 ** -0.5).to(dt)
b1 = torch.randn(2560, device=dev)
cs = w1.float().sum(1).contiguous()
K.attach_w_frag(w1, True)
w2 = (torch.randn(320, 1280, device=dev) * 1280 ** -0.5).to(dt)
b2 = torch.randn(320, device=dev)
h = torch.empty(m, 1280, device=dev, dtype=dt)
y = torch.empty(m, 320, device=dev, dtype=dt)
junk = [torch.empty(64 << 20, device=dev, dtype=torch.uint8) for _ in range(2)]


def pair(ev):
    K.gemm(x, w1, bias=b1, geglu=True, ln=(K.RowStats(x, 1e-5), cs), out=h)
    ev[0].record()
    K.gemm(h, w2, bias=b2, residual=x, out=y)
    ev[1].record()


evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(30)]
for it in range(30):
    evs[it][2].record()
    pair(evs[it])
torch.cuda.synchronize()
prod = sorted(e[2].elapsed_time(e[0]) * 1e3 for e in evs[5:])
cons = sorted(e[0].elapsed_time(e[1]) * 1e3 for e in evs[5:])
print(f"CA_PP_DBG={os.environ.get('CA_PP_DBG', '0')}: GEGLU 131072x2560x320 {prod[len(prod) // 2]:.1f} us, FF-out 131072x320x1280 (+residual) {cons[len(cons) // 2]:.1f} us (medians of 25)")
