import sys; sys.path.insert(0, "/root/repo")
import torch
from controlanimate_amd import kernels as K
m, n, k = [int(x) for x in sys.argv[1].split("x")]
a = torch.randn(m, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
for _ in range(2): K.gemm(a, w)
torch.cuda.synchronize()
