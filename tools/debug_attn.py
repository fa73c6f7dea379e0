"""Debug helper: compares ca_attention (spatial self-attention) with a torch fp32 reference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlanimate_amd import kernels as K
images, heads, d, n = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (1, 1, 40, 256)))
dt = torch.float16 if (len(sys.argv) < 6 or sys.argv[5] == "f16") else torch.bfloat16
c = heads * d
torch.manual_seed(0)
qkv = torch.randn(images * n, 3 * c, device="cuda").to(dt)
o = K.attention_spatial(qkv, images, n, heads).float()
x = qkv.float().view(images, n, 3, heads, d).permute(2, 0, 3, 1, 4)
ref = torch.softmax(x[0] @ x[1].transpose(-1, -2) * d ** -0.5, -1) @ x[2]
ref = ref.permute(0, 2, 1, 3).reshape(images * n, c)
bad = ~torch.isfinite(o)
print("nonfinite:", bad.sum().item(), "of", o.numel())
if bad.any():
    idx = bad.nonzero()
    print("rows:", idx[:, 0].unique()[:32].tolist(), "cols:", idx[:, 1].unique()[:48].tolist())
print("rel err (finite part):", ((o - ref)[~bad].norm() / ref[~bad].norm()).item())
