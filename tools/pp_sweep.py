import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlanimate_amd import kernels as K
from tools.pp_check import timeit
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("CA_"))
shapes = [(8192, 10240, k) for k in (640, 1280, 2560, 5120, 10240)] + [(8192, 8192, 8192), (16384, 4096, 1280), (65536, 1280, 1280)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for (m, n, k) in shapes:
    a = torch.randn(m, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
    ms = timeit(lambda: K.gemm(a, w), it=5)
    print(f"gemm {m}x{n}x{k}: {ms*1e3:8.1f} us {2.0*m*n*k/ms/1e9:7.1f} TF  [{tag}]", flush=True)
