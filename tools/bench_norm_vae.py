"""GroupNorm + SiLU at the SD1.5 VAE's shapes (16 frames of a 512 x 512 window): the two-kernel path pass by pass against a plain copy.
    python tools/bench_norm_vae.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from controlanimate_amd import _capi
from controlanimate_amd import kernels as K
from controlanimate_amd.kernels import _p, _stream, dt_code
from tools.bench_norm import timeit
lib = _capi.lib()
for (img, h, c) in [(16, 512, 128), (16, 256, 256), (16, 256, 128), (16, 128, 512), (16, 128, 256), (16, 64, 512)]:
    x = torch.randn(img, h, h, c, device="cuda").half(); g = torch.ones(c, device="cuda"); b = torch.zeros(c, device="cuda")
    y = torch.empty_like(x)
    nfl = lib.ca_groupnorm_partials_floats(img, h * h, 1, 32)
    partials = torch.empty((max(int(nfl), 1),), device="cuda", dtype=torch.float32)
    args = _capi.GroupNormArgs(x=_p(x), x2=None, y=_p(y), gamma=_p(g), beta=_p(b), partials=_p(partials), images=img, hw=h * h, c1=c, c2=0, groups=32,
                               frames_per_stat=1, eps=1e-6, act=1, dtype=dt_code(x.dtype))
    one = timeit(lambda: _capi.check(lib.ca_groupnorm(C.byref(args), _stream()), "gn"))
    st = timeit(lambda: _capi.check(lib.ca_groupnorm_stats(C.byref(args), _stream()), "gn_stats"))
    ap = timeit(lambda: _capi.check(lib.ca_groupnorm_apply(C.byref(args), _stream()), "gn_apply"))
    cp = timeit(lambda: y.copy_(x))
    gb = x.numel() * 2 / 1e9
    print(f"groupnorm+silu {img}x{h}x{h}x{c} ({gb:.2f} GB): ca_groupnorm {one*1e3:7.1f} us | stats pass {st*1e3:7.1f} ({gb/st:5.2f} TB/s) | apply pass {ap*1e3:7.1f} ({2*gb/ap:5.2f} TB/s) | plain copy {cp*1e3:7.1f} ({2*gb/cp:5.2f} TB/s)", flush=True)
