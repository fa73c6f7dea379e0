"""How much the matrix pipes' rate depends on the operand DATA (MI355X issues and clocks to its power budget): the 256 x 320
kernel and torch.matmul (hipBLASLt; a measurement reference only) on N(0,1), zero, constant, coarse (few mantissa bits) and
random-bit operands.  Round-3 result: DESIGN.md section 3.     python tools/gemm_data_power.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlanimate_amd import kernels as K
DEV = "cuda"
def timeit(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
def data(mode, *shape, scale=1.0):
    if mode == "zeros": return torch.zeros(*shape, device=DEV).half()
    if mode == "ones": return torch.full(shape, 1.0, device=DEV).half()
    x = torch.randn(*shape, device=DEV) * scale
    if mode == "randn": return x.half()
    if mode == "coarse": return (torch.round(x * 4) / 4).half()   # few mantissa bits set
    if mode == "bits": return torch.randint(0, 0x3BFF, shape, device=DEV, dtype=torch.int16).view(torch.float16)
for (m, n, k) in [(32768, 640, 2560), (8192, 8192, 8192)]:
    for mode in ("randn", "zeros", "ones", "coarse", "bits", "randn"):
        a = data(mode, m, k); w = data(mode, n, k, scale=k ** -0.5)
        row = f"{m}x{n}x{k} {mode:6s}"
        if n % 320 == 0:
            K._plan_sink = lab = []; K.gemm(a, w); K._plan_sink = None
            ms = timeit(lambda: K.gemm(a, w))
            row += f" | ours {lab[0]} {ms*1e3:7.1f}us {2.0*m*n*k/ms/1e9:6.0f}TF"
        ms = timeit(lambda: torch.matmul(a, w.t()))
        row += f" | torch.matmul {ms*1e3:7.1f}us {2.0*m*n*k/ms/1e9:6.0f}TF"
        print(row, flush=True)
