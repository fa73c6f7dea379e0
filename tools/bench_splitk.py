"""Split-K sweep helper for the 8x8 / 16x16-latent convolutions: CA_SPLITK=<S> python tools/bench_splitk.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlanimate_amd import kernels as K
from tools.bench_gemm import timeit  # noqa: E402  (prints its own table first)
tag = f"SPLITK={os.environ.get('CA_SPLITK','auto')}"
for (img, h, ci, co) in [(32, 8, 1280, 1280), (32, 8, 2560, 1280), (32, 16, 1280, 1280), (32, 16, 2560, 1280)]:
    x = torch.randn(img, h, h, ci, device="cuda").half(); w = (torch.randn(co, 3, 3, ci, device="cuda") * (9 * ci) ** -0.5).half()
    ms = timeit(lambda: K.conv3x3(x, w))
    print(f"conv {img}x{h}x{h} {ci}->{co}: {ms*1e3:8.1f} us {2.0*img*h*h*co*9*ci/ms/1e9:7.1f} TF  [{tag}]")
