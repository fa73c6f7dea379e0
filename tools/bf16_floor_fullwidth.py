"""The bf16 operand-rounding floor of UNet eps (tests/test_bf16_floor_cpu.py) at FULL SD1.5 width: the fp32 oracle re-run with
only the matrix-multiply operands rounded to bf16 / fp16, on the BASELINE config-1 shape (mm v1, 8 frames, 32x32 latents, CFG
batch 2) and on a v2 model at the same shape, seeded random weights (zero-initialised projections re-randomised as bench.py does).
CPU only (about a minute per forward on 8 cores):
    python tools/bf16_floor_fullwidth.py > profiles/round4_bf16_floor_fullwidth.txt
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_bf16_floor_cpu import _emulated
from oracle.unet3d import UNet3DConfig, init_unet3d_weights, unet3d_forward

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
print(f"# fp32 oracle, full width (320, 640, 1280, 1280), {_cores()} threads; relative L2 error of eps against the exact fp32 run")
for name, cfg, f in (("mm v1 (BASELINE config 1 shape: 8 frames, 32x32 latents)", UNet3DConfig.v1(), 8), ("mm v2, 8 frames, 32x32 latents", UNet3DConfig.v2(), 8)):
    w = init_unet3d_weights(cfg, seed=5)
    g = torch.Generator().manual_seed(7)
    for k, v in w.items():  # real checkpoints are non-zero where the architecture zero-initialises (motion proj_out)
        if v.dim() > 1 and float(v.abs().max()) == 0.0:
            w[k] = torch.randn(v.shape, generator=g) * 0.02
    args = (torch.randn(2, 4, f, 32, 32, generator=g), 500, torch.randn(2, 77, 768, generator=g) * 0.5)
    with torch.no_grad():
        t0 = time.time()
        ref = unet3d_forward(w, cfg, *args)
        t1 = time.time() - t0
        for dt in (torch.bfloat16, torch.float16):
            out = _emulated(dt, w, cfg, args)
            print(f"{name}: {str(dt)[6:]:9s} operands -> {((out - ref).norm() / ref.norm()).item():.3e}   (forward {t1:.0f} s)", flush=True)
