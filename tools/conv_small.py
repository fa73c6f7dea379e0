import os, sys
sys.path.insert(0, "/root/repo")
import torch, torch.nn.functional as F
from controlanimate_amd import kernels as K
torch.manual_seed(0)
for (img, h, ci, co) in [(8, 32, 320, 320), (32, 32, 320, 320), (32, 64, 320, 320)]:
    x = torch.randn(img, h, h, ci, device="cuda").half(); w = (torch.randn(co, 3, 3, ci, device="cuda") * (9 * ci) ** -0.5).half()
    y = K.conv3x3(x, w); torch.cuda.synchronize()
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    print(img, h, ci, co, "rel", ((y.float() - ref).norm() / ref.norm()).item(), flush=True)
