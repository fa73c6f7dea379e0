"""UNet3D eps of the HIP path in fp16 and bf16 at FULL SD1.5 width against the fp32 oracle (host cores), mm v1 and mm v2, on the
BASELINE config-1 shape -- beside the operand-rounding floor of tools/bf16_floor_fullwidth.py.
    python tools/bf16_fullwidth_gpu.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlanimate_amd.configs import unet_config
from controlanimate_amd.unet import UNet3DConditionModel
from oracle.unet3d import UNet3DConfig, init_unet3d_weights, unet3d_forward

DEV = "cuda:0"
def _cores():  # (the cgroup's share, not the machine's core count: bench.usable_cores)
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 32))


torch.set_num_threads(_cores())
for ver, cfg in (("v1", UNet3DConfig.v1()), ("v2", UNet3DConfig.v2())):
    w = init_unet3d_weights(cfg, seed=5)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 4, 8, 32, 32, generator=g)
    ehs = torch.randn(2, 77, 768, generator=g) * 0.5
    with torch.no_grad():
        ref = unet3d_forward(w, cfg, x, 500, ehs)
    for dt in (torch.float16, torch.bfloat16):
        m = UNet3DConditionModel.from_config(unet_config(ver))
        m.load_state_dict(w, strict=False)
        m.to(DEV).prepare(DEV, dt)
        out = m(x.to(DEV), 500, ehs.to(DEV)).sample.float().cpu()
        print(f"mm {ver} full width (2,4,8,32,32): HIP {str(dt)[6:]:9s} eps rel_l2 vs fp32 oracle = {((out - ref).norm() / ref.norm()).item():.3e}", flush=True)
        del m
        torch.cuda.empty_cache()
