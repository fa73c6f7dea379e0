"""One VAE decode + encode of a 16-frame 512x512 window (for rocprofv3): python tools/vae_one.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlanimate_amd.vae import AutoencoderKL
torch.manual_seed(0)
vae = AutoencoderKL.from_config().to("cuda").prepare("cuda", torch.float16)
lat = torch.randn(16, 4, 64, 64, device="cuda")
img = torch.rand(16, 3, 512, 512, device="cuda") * 2 - 1
for _ in range(2):
    vae.decode(lat)
    vae.encode_moments(img)
torch.cuda.synchronize()
