"""LayerNorm -> GEMM versus row statistics + LayerNorm-folded GEMM at the UNet's transformer shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlanimate_amd import kernels as K
from tools.bench_gemm import timeit
for (m, n, k, geglu) in [(131072, 960, 320, False), (131072, 320, 320, False), (131072, 2560, 320, True), (32768, 1920, 640, False), (32768, 5120, 640, True),
                         (8192, 3840, 1280, False), (8192, 10240, 1280, True), (2048, 3840, 1280, False), (2048, 10240, 1280, True)]:
    x = torch.randn(m, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
    g = torch.ones(k, device="cuda"); b = torch.zeros(k, device="cuda"); bias = torch.zeros(n, device="cuda"); cs = w.float().sum(1)
    t_ln = timeit(lambda: K.layer_norm(x, g, b))
    xn = K.layer_norm(x, g, b)
    t_g = timeit(lambda: K.gemm(xn, w, bias=bias, geglu=geglu))
    t_st = timeit(lambda: K.row_stats(x))
    st = K.row_stats(x)
    t_gl = timeit(lambda: K.gemm(x, w, bias=bias, geglu=geglu, ln=(st, cs)))
    t_a = timeit(lambda: K.gemm(K.layer_norm(x, g, b), w, bias=bias, geglu=geglu))
    t_b = timeit(lambda: K.gemm(x, w, bias=bias, geglu=geglu, ln=(K.row_stats(x), cs)))
    K._plan_sink = lab = []
    K.gemm(xn, w, bias=bias, geglu=geglu); K.gemm(x, w, bias=bias, geglu=geglu, ln=(st, cs))
    K._plan_sink = None
    t_blas = timeit(lambda: torch.nn.functional.linear(xn, w, bias.half()))
    print(f"   kernels: plain {lab[0]}, folded {lab[1]}; vendor BLAS on the normalised input (+ bias) {t_blas*1e3:.1f} us")
    print(f"{m}x{n}x{k}{' geglu' if geglu else ''}: LN {t_ln*1e3:.1f} + GEMM {t_g*1e3:.1f} (chained {t_a*1e3:.1f}) | stats {t_st*1e3:.1f} + GEMM(ln) {t_gl*1e3:.1f} (chained {t_b*1e3:.1f}) us")
