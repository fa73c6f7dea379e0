"""Correctness + timing of the GEMM / conv kernels at UNet shapes under the current CA_GEMM_* environment.
    CA_GEMM_PP=1 python tools/pp_check.py [--check] [--time]
Correctness: vs a torch fp32 matmul / conv2d of the same fp16 operands; every launch is repeated and compared
bit-for-bit with the first (a race shows up as run-to-run differences)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from controlanimate_amd import kernels as K

def timeit(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it

GEMMS = [(131072, 2560, 320), (131072, 320, 320), (32768, 5120, 640), (32768, 640, 640), (8192, 10240, 1280), (8192, 1280, 1280),
         (131072, 960, 320), (131072, 320, 1280), (32768, 1920, 640), (8192, 3840, 1280), (32768, 640, 2560), (8192, 1280, 5120),
         (2048, 1280, 1280), (2048, 10240, 1280), (2048, 1280, 5120)]
CONVS = [(32, 64, 320, 320), (32, 32, 640, 640), (32, 16, 1280, 1280), (32, 8, 1280, 1280), (32, 64, 640, 320), (32, 32, 1280, 640), (32, 16, 2560, 1280)]

def check():
    dev = "cuda"
    torch.manual_seed(0)
    bad = 0
    def report(name, out, ref, reps):
        nonlocal bad
        rel = ((out.float() - ref).norm() / ref.norm()).item()
        same = all(torch.equal(out, r) for r in reps)
        flag = "" if (rel < 4e-3 and same) else "   <<<<<< FAIL"
        if flag: bad += 1
        print(f"{name}: rel {rel:.2e} deterministic={same}{flag}", flush=True)
    for dt in (torch.float16, torch.bfloat16):
        tol_name = "f16" if dt == torch.float16 else "bf16"
        for (m, n, k) in [(8192, 1280, 1280), (8192 - 40, 2560, 640), (1000, 1280, 320), (4096, 256, 128), (2048, 10240, 1280), (32768, 640, 2560), (512, 384, 192)]:
            a = torch.randn(m, k, device=dev).to(dt); w = (torch.randn(n, k, device=dev) * k ** -0.5).to(dt)
            bias = torch.randn(n, device=dev); res = torch.randn(m, n, device=dev).to(dt)
            ref = a.float() @ w.float().t()
            outs = [K.gemm(a, w) for _ in range(4)]
            report(f"gemm {tol_name} {m}x{n}x{k}", outs[0], ref, outs[1:])
            ref2 = (a.float() @ w.float().t() + bias).to(dt).float() + res.float()
            outs = [K.gemm(a, w, bias=bias, residual=res) for _ in range(3)]
            report(f"gemm+bias+res {tol_name} {m}x{n}x{k}", outs[0], ref2, outs[1:])
        # GEGLU epilogue + two-source A
        m, n, k = 4096, 2560, 640
        a1 = torch.randn(m, 384, device=dev).to(dt); a2 = torch.randn(m, 256, device=dev).to(dt)
        w = (torch.randn(n, k, device=dev) * k ** -0.5).to(dt)
        y = torch.cat([a1, a2], 1).float() @ w.float().t()
        ref = y[:, 0::2] * F.gelu(y[:, 1::2])
        outs = [K.gemm(a1, w, a2=a2, geglu=True) for _ in range(3)]
        report(f"gemm geglu 2-src {tol_name}", outs[0], ref, outs[1:])
        for (img, h, ci, co, stride, ups, c2) in [(4, 32, 640, 1280, 1, 0, 0), (4, 32, 640, 1280, 2, 0, 0), (4, 16, 1280, 1280, 1, 1, 0), (2, 32, 640, 512, 1, 0, 640), (3, 30, 320, 256, 1, 0, 0)]:
            x = torch.randn(img, h, h, ci, device=dev).to(dt)
            x2 = torch.randn(img, h, h, c2, device=dev).to(dt) if c2 else None
            w = (torch.randn(co, 3, 3, ci + c2, device=dev) * (9 * (ci + c2)) ** -0.5).to(dt)
            xin = torch.cat([x, x2], 3) if c2 else x
            xn = xin.float().permute(0, 3, 1, 2)
            if ups: xn = F.interpolate(xn, scale_factor=2, mode="nearest")
            ref = F.conv2d(xn, w.float().permute(0, 3, 1, 2), stride=stride, padding=1).permute(0, 2, 3, 1)
            outs = [K.conv3x3(x, w, x2=x2, stride=stride, upsample=ups) for _ in range(3)]
            report(f"conv {tol_name} {img}x{h}x{h} {ci}+{c2}->{co} s{stride} u{ups}", outs[0], ref, outs[1:])
    print("FAILURES:", bad)

def time_all():
    tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("CA_GEMM"))
    tot = 0.0
    for (m, n, k) in GEMMS:
        a = torch.randn(m, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
        ms = timeit(lambda: K.gemm(a, w))
        print(f"gemm {m}x{n}x{k}: {ms*1e3:8.1f} us {2.0*m*n*k/ms/1e9:7.1f} TF  [{tag}]", flush=True)
    for (img, h, ci, co) in CONVS:
        x = torch.randn(img, h, h, ci, device="cuda").half(); w = (torch.randn(co, 3, 3, ci, device="cuda") * (9 * ci) ** -0.5).half()
        ms = timeit(lambda: K.conv3x3(x, w))
        print(f"conv {img}x{h}x{h} {ci}->{co}: {ms*1e3:8.1f} us {2.0*img*h*h*co*9*ci/ms/1e9:7.1f} TF  [{tag}]", flush=True)

if __name__ == "__main__":
    if "--check" in sys.argv: check()
    if "--time" in sys.argv: time_all()
