"""s_memtime stamps of one block of the persistent streaming kernel (stamps library: python -m controlanimate_amd._build --experiments --stamps;
CA_HIP_LIB=controlanimate_amd/csrc/libcontrolanimate_hip_stamps.so CA_PP_DBG=9 CA_GEMM_PS=1):
    python tools/ps_stamps.py M N K [geglu|res]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlanimate_amd import kernels as K
m, n, k = (int(x) for x in sys.argv[1:4])
extra = sys.argv[4] if len(sys.argv) > 4 else ""
a = torch.randn(m, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
kw = dict(bias=torch.randn(n, device="cuda"))
if extra == "geglu": kw["geglu"] = True
if extra == "res": kw["residual"] = torch.randn(m, n, device="cuda").half()
K.gemm(a, w, **kw)
torch.cuda.synchronize()
