"""s_memtime stamps of one block of the 256 x 320 streaming kernel (stamps library: python -m controlanimate_amd._build --experiments --stamps;
CA_HIP_LIB=controlanimate_amd/csrc/libcontrolanimate_hip_stamps.so CA_PP_DBG=9 CA_GEMM_PQ=1; the stamp code costs registers: the kernel spills with it):
    python tools/pq_stamps.py conv IMAGES H CIN COUT | gemm M N K [res]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlanimate_amd import kernels as K
if sys.argv[1] == "conv":
    img, h, ci, co = (int(x) for x in sys.argv[2:6])
    x = torch.randn(img, h, h, ci, device="cuda").half(); w = (torch.randn(co, 3, 3, ci, device="cuda") * (9 * ci) ** -0.5).half()
    K.conv3x3(x, w, bias=torch.randn(co, device="cuda"))
else:
    m, n, k = (int(x) for x in sys.argv[2:5])
    a = torch.randn(m, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
    kw = dict(bias=torch.randn(n, device="cuda"))
    if len(sys.argv) > 5 and sys.argv[5] == "res": kw["residual"] = torch.randn(m, n, device="cuda").half()
    K.gemm(a, w, **kw)
torch.cuda.synchronize()
