"""What the epilogue of the 256 x 320 streaming kernel costs per launch: the experiments library's CA_PP_DBG switch (1 = no epilogue,
2 = no main loop; timing only) on the wide-N shapes of the step, plain (bias) and LayerNorm-folded (+ GEGLU) epilogues.

    python -m controlanimate_amd._build --experiments
    for d in 0 1 2; do CA_HIP_LIB=$PWD/controlanimate_amd/csrc/libcontrolanimate_hip_exp.so CA_PP_DBG=$d python tools/pq_epilogue_cost.py; done
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlanimate_amd import kernels as K
from tools.bench_gemm import timeit
tag = "CA_PP_DBG=" + os.environ.get("CA_PP_DBG", "0")
out = []
for (m, n, k, geglu) in [(8192, 10240, 1280, True), (32768, 5120, 640, True), (8192, 3840, 1280, False), (32768, 1920, 640, False)]:
    x = torch.randn(m, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
    bias = torch.zeros(n, device="cuda"); cs = w.float().sum(1)
    st = K.row_stats(x)
    K._plan_sink = lab = []
    K.gemm(x, w, bias=bias); K.gemm(x, w, bias=bias, geglu=geglu, ln=(st, cs))
    K._plan_sink = None
    t_p = timeit(lambda: K.gemm(x, w, bias=bias))
    t_l = timeit(lambda: K.gemm(x, w, bias=bias, geglu=geglu, ln=(st, cs)))
    out.append(f"{m}x{n}x{k}: plain {t_p*1e3:6.1f} ({lab[0]}) ln{'+geglu' if geglu else ''} {t_l*1e3:6.1f} ({lab[1]})")
print(f"{tag:12s} | " + " | ".join(out), flush=True)
