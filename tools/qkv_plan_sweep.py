"""Which kernel should run the LayerNorm-folded q|k|v projections of the 16x16- and 32x32-latent levels?  Experiments library, plan knobs from the
environment (CA_GEMM_PS / CA_GEMM_PQ / CA_GEMM_PP); prints the plan label and the time per shape.
    for k in "" CA_GEMM_PS=1 CA_GEMM_PQ=1 CA_GEMM_PP=2; do env $k CA_HIP_LIB=$PWD/controlanimate_amd/csrc/libcontrolanimate_hip_exp.so python tools/qkv_plan_sweep.py; done"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlanimate_amd import kernels as K
from tools.bench_gemm import timeit
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("CA_GEMM")) or "default"
out = []
for (m, n, k) in [(8192, 3840, 1280), (32768, 1920, 640), (2048, 3840, 1280), (4096, 3840, 1280), (16384, 1920, 640)]:
    x = torch.randn(m, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
    bias = torch.zeros(n, device="cuda"); cs = w.float().sum(1)
    st = K.row_stats(x)
    K._plan_sink = lab = []
    K.gemm(x, w, bias=bias, ln=(st, cs))
    K._plan_sink = None
    t = timeit(lambda: K.gemm(x, w, bias=bias, ln=(st, cs)), it=30)
    out.append(f"{m}x{n}x{k}: {t*1e3:6.1f} us {2.0*m*n*k/t/1e9:5.0f} TF ({lab[0]})")
print(f"{tag:16s} | " + " | ".join(out), flush=True)
