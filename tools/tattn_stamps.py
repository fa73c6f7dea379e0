"""Shader-clock stamps of the fused temporal attention (experiments build): block 0, wave 0, its tiles.
    CA_HIP_LIB=.../libcontrolanimate_hip_exp.so python tools/tattn_stamps.py
tags: 0 tile start, 1 x tile landed, 2 LayerNorm done, per head: 3 q loop, 4 q packed, 5 k loop, 6 softmax done, 7 v loop, 8 stores issued"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import tattn_check as T
from controlanimate_amd import kernels as K
from controlanimate_amd.layers import frag_order_tattn

x, w, gamma, beta, pe = T.make(2, 4096, torch.float16)
wl = frag_order_tattn(w.float()).to(torch.float16)
bp = (pe + beta[None, :]).contiguous()
for _ in range(3):
    K.tattn_fused(x, wl, gamma, bp, 2, 16, 4096, 8, 1e-5, 40 ** -0.5)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 512)()
lib = K.lib()
lib.ca_debug_ff_stamps.restype = C.c_int
assert lib.ca_debug_ff_stamps(buf) == 0
st = [(buf[2 * i], buf[2 * i + 1]) for i in range(120) if buf[2 * i]]
prev = st[0][0]
line = []
for t, tag in st:
    if tag == 0 and line:
        print("  ", " ".join(line))
        line = []
    line.append(f"{tag}:{t - prev}")
    prev = t
print("  ", " ".join(line))
print("   total", st[-1][0] - st[0][0])
