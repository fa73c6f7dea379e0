"""Micro-benchmark of GroupNorm / LayerNorm at UNet shapes.  python tools/bench_norm.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlanimate_amd import kernels as K
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
for (img, h, c) in [(32, 64, 320), (32, 32, 640), (32, 16, 1280), (32, 8, 1280), (32, 64, 960), (32, 32, 1920)]:
    x = torch.randn(img, h, h, c, device="cuda").half(); g = torch.ones(c, device="cuda"); b = torch.zeros(c, device="cuda")
    ms = timeit(lambda: K.group_norm(x, g, b, act=1))
    print(f"groupnorm {img}x{h}x{h}x{c}: {ms*1e3:7.1f} us  {3*x.numel()*2/ms/1e9:6.2f} TB/s (2 reads + 1 write)")
for (rows, c) in [(131072, 320), (32768, 640), (8192, 1280)]:
    x = torch.randn(rows, c, device="cuda").half(); g = torch.ones(c, device="cuda"); b = torch.zeros(c, device="cuda")
    ms = timeit(lambda: K.layer_norm(x, g, b))
    print(f"layernorm {rows}x{c}: {ms*1e3:7.1f} us  {2*x.numel()*2/ms/1e9:6.2f} TB/s")

# ---- GroupNorm + SiLU: the one-launch kernel against an APPLY-ONLY pass (statistics already known) -- the floor of any design that
# takes the statistics from the producing convolution's epilogue and still materialises the normalised tensor once
import ctypes as C
from controlanimate_amd import _capi
from controlanimate_amd.kernels import _p, _stream, dt_code
lib = _capi.lib()
for (img, h, c) in [(32, 64, 320), (32, 32, 640), (32, 16, 1280), (16, 64, 320)]:
    x = torch.randn(img, h, h, c, device="cuda").half(); g = torch.ones(c, device="cuda"); b = torch.zeros(c, device="cuda")
    y = torch.empty_like(x)
    nfl = lib.ca_groupnorm_partials_floats(img, h * h, 1, 32)
    partials = torch.empty((max(int(nfl), 1),), device="cuda", dtype=torch.float32)
    args = _capi.GroupNormArgs(x=_p(x), x2=None, y=_p(y), gamma=_p(g), beta=_p(b), partials=_p(partials), images=img, hw=h * h, c1=c, c2=0, groups=32,
                               frames_per_stat=1, eps=1e-5, act=1, dtype=dt_code(x.dtype))
    one = timeit(lambda: _capi.check(lib.ca_groupnorm(C.byref(args), _stream()), "gn"))
    st = timeit(lambda: _capi.check(lib.ca_groupnorm_stats(C.byref(args), _stream()), "gn_stats"))
    ap = timeit(lambda: _capi.check(lib.ca_groupnorm_apply(C.byref(args), _stream()), "gn_apply"))
    cp = timeit(lambda: y.copy_(x))
    print(f"groupnorm+silu {img}x{h}x{h}x{c}: one launch {one*1e3:6.1f} us | stats pass {st*1e3:6.1f} | apply-only pass {ap*1e3:6.1f} | plain copy {cp*1e3:6.1f}")
