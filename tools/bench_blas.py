"""What the vendor BLAS reaches on the UNet's GEMM shapes (torch.matmul -> hipBLASLt/rocBLAS), next to ca_gemm.
Measurement aid only: the product never calls a BLAS library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlanimate_amd import kernels as K
from tools.bench_gemm import timeit
for (m, n, k) in [(131072, 320, 320), (131072, 2560, 320), (131072, 960, 320), (32768, 5120, 640), (32768, 640, 640), (8192, 10240, 1280),
                  (131072, 320, 1280), (8192, 1280, 5120), (8192, 1280, 1280), (32768, 640, 2560), (8192, 1280, 11520), (2048, 1280, 11520), (32768, 640, 5760),
                  (8192, 3840, 1280), (32768, 1920, 640), (2048, 1280, 1280), (2048, 3840, 1280), (2048, 1280, 5120)]:
    a = torch.randn(m, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
    wt = w.t()
    ms_b = timeit(lambda: torch.matmul(a, wt))
    ms_c = timeit(lambda: K.gemm(a, w))
    print(f"{m:7d}x{n:5d}x{k:5d}: BLAS {ms_b*1e3:8.1f} us {2.0*m*n*k/ms_b/1e9:7.1f} TF | ca_gemm {ms_c*1e3:8.1f} us {2.0*m*n*k/ms_c/1e9:7.1f} TF")
