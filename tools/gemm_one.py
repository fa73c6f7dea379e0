"""Runs one GEMM / conv shape a few times (for rocprofv3 --pmc passes): python tools/gemm_one.py gemm M N K | conv IMG H CIN COUT"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlanimate_amd import kernels as K
kind = sys.argv[1]
a, b, c, d = (int(x) for x in sys.argv[2:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
if kind == "gemm":
    x = torch.randn(a, c, device="cuda").half(); w = (torch.randn(b, c, device="cuda") * c ** -0.5).half()
    fn = lambda: K.gemm(x, w)
else:
    x = torch.randn(a, b, b, c, device="cuda").half(); w = (torch.randn(d, 3, 3, c, device="cuda") * (9 * c) ** -0.5).half()
    fn = lambda: K.conv3x3(x, w)
for _ in range(reps):
    fn()
torch.cuda.synchronize()
