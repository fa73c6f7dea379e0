"""Micro-benchmark of ca_gemm / ca_conv3x3 at UNet shapes.  python tools/bench_gemm.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlanimate_amd import kernels as K

def timeit(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it

if __name__ == "__main__":
    tag = f"NBUF={os.environ.get('CA_GEMM_NBUF','-')} BN={os.environ.get('CA_GEMM_BN','-')}"
    for (m, n, k) in [(131072, 320, 320), (131072, 2560, 320), (32768, 5120, 640), (8192, 10240, 1280), (131072, 320, 1280), (8192, 1280, 5120), (8192, 1280, 1280)]:
        a = torch.randn(m, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
        ms = timeit(lambda: K.gemm(a, w))
        print(f"gemm {m}x{n}x{k}: {ms*1e3:8.1f} us {2.0*m*n*k/ms/1e9:7.1f} TF  [{tag}]")
    for (img, h, ci, co) in [(32, 64, 320, 320), (32, 32, 640, 640), (32, 16, 1280, 1280), (32, 8, 1280, 1280)]:
        x = torch.randn(img, h, h, ci, device="cuda").half(); w = (torch.randn(co, 3, 3, ci, device="cuda") * (9 * ci) ** -0.5).half()
        ms = timeit(lambda: K.conv3x3(x, w))
        print(f"conv {img}x{h}x{h} {ci}->{co}: {ms*1e3:8.1f} us {2.0*img*h*h*co*9*ci/ms/1e9:7.1f} TF  [{tag}]")
