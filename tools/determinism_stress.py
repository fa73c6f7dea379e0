"""Repeat a full-width UNet3D forward (config-4 geometry: 16 frames, 64x96 latents, CFG batch 2) and count results that
differ from the first one, bit for bit -- a race shows up as a non-zero count.  Tuning / kernel-selection environment
variables apply (one process per setting):  CA_GEMM_WRES=0 python tools/determinism_stress.py [iterations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_workload_configs_gpu import _full_unet, DEV
from controlanimate_amd import kernels as K
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
f, h, w = 16, 64, 96
unet = _full_unet()
g = torch.Generator().manual_seed(5)
lat = torch.randn(1, 4, f, h, w, generator=g).to(DEV)
pos = (torch.randn(1, 77, 768, generator=g) * 0.5).to(DEV)
same = torch.cat([pos, pos]).contiguous()
x2 = K.latents_to_nhwc(lat, unet.conv_in.cin_pad, 2, 1.0, torch.float16)
ref = unet.forward_nhwc(x2, 2, f, 500, same).clone()
bad = halves = 0
for i in range(n):
    e = unet.forward_nhwc(x2, 2, f, 500, same)
    torch.cuda.synchronize()
    bad += int(not torch.equal(e, ref))
    halves += int(not torch.equal(e[:f], e[f:]))
print(f"{n} forwards: {bad} differ from the first, {halves} with unequal CFG halves   env: " +
      " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("CA_")))
