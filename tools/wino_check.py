"""Winograd F(2x2, 3x3) route of ca_conv3x3 (csrc/ca_conv_wino.h): correctness against fp32 torch and against the direct implicit-GEMM
form, the library's weight transform against layers.HipConv3x3._winograd_weight, determinism and timing -- one process.
    python tools/wino_check.py [--time-only]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from controlanimate_amd import kernels as K

dev = "cuda"
G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]])


def make(images, h, c1, c2, cout, dt=torch.float16, seed=3, epilogue=True):
    g = torch.Generator(device="cpu").manual_seed(seed)
    rn = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev)
    x = rn(images, h, h, c1).to(dt)
    x2 = rn(images, h, h, c2).to(dt) if c2 else None
    w = rn(cout, c1 + c2, 3, 3, scale=(9 * (c1 + c2)) ** -0.5)          # fp32 [Cout, Cin, kh, kw]
    d = dict(x=x, x2=x2, w32=w, w=w.permute(0, 2, 3, 1).contiguous().to(dt), u=torch.einsum("xk,oikl,yl->xyoi", G.to(dev), w, G.to(dev)).reshape(16, cout, c1 + c2).contiguous().to(dt),
             bias=None, rowbias=None, residual=None, rows_per_group=0, post=1.0)
    if epilogue:
        d.update(bias=rn(cout, scale=0.1), rowbias=rn(2, cout, scale=0.3), residual=rn(images, h, h, cout).to(dt), rows_per_group=images // 2 * h * h, post=1.0 / 1.3)
    return d


def reference(d):
    x = d["x"].float() if d["x2"] is None else torch.cat([d["x"].float(), d["x2"].float()], dim=-1)
    y = F.conv2d(x.permute(0, 3, 1, 2), d["w"].float().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)  # (the ROUNDED weights both forms start from)
    if d["bias"] is not None:
        y = y + d["bias"]
    if d["rowbias"] is not None:
        imgs = y.shape[0]
        y = y + d["rowbias"].repeat_interleave(imgs // 2, dim=0)[:, None, None, :]
    if d["residual"] is not None:
        y = y + d["residual"].float()
    return y * d["post"]


def run(d, winograd):
    return K.conv3x3(d["x"], d["w"], x2=d["x2"], bias=d["bias"], rowbias=d["rowbias"], rows_per_group=d["rows_per_group"], residual=d["residual"],
                     post_scale=d["post"], w_wino=d["u"] if winograd else None)


SHAPES = [(32, 16, 1280, 0, 1280), (32, 16, 1280, 1280, 1280), (32, 16, 1280, 640, 1280), (32, 8, 1280, 0, 1280), (32, 8, 1280, 1280, 1280), (16, 16, 1280, 640, 640)]


def check():
    bad = 0
    for (images, h, c1, c2, cout) in SHAPES:
        for epi in (True, False):
            d = make(images, h, c1, c2, cout, epilogue=epi)
            K._plan_sink = labels = []
            try:
                outs = [run(d, True) for _ in range(2)]
            finally:
                K._plan_sink = None
            direct = run(d, False)
            ref = reference(d)
            rel = ((outs[0].float() - ref).norm() / ref.norm()).item()
            rel_d = ((direct.float() - ref).norm() / ref.norm()).item()
            mx = (outs[0].float() - ref).abs().max().item()
            ok = labels[0].startswith("wino") and rel < 3e-3 and torch.equal(outs[0], outs[1]) and bool(torch.isfinite(outs[0].float()).all())
            bad += not ok
            print(f"images={images} {h}x{h} cin={c1}+{c2} cout={cout} epilogue={epi}: {labels[0]}  rel {rel:.2e} (direct {rel_d:.2e})  max abs {mx:.2e}"
                  f"{'' if ok else '   <<<<<< FAIL'}", flush=True)
    # the library's weight transform (from the 16-bit weights) against torch
    d = make(4, 8, 1280, 0, 320)
    u = torch.empty(16, 320, 1280, device=dev, dtype=torch.float16)
    K.check(K.lib().ca_pack_w_wino(d["w"].data_ptr(), 320, 1280, K.dt_code(torch.float16), u.data_ptr(), K._stream()), "ca_pack_w_wino")
    want = torch.einsum("xk,oikl,yl->xyoi", G.to(dev), d["w"].float().permute(0, 3, 1, 2), G.to(dev)).reshape(16, 320, 1280)
    err = (u.float() - want).abs().max().item()
    print(f"ca_pack_w_wino vs torch: max abs {err:.2e}{'' if err < 2e-3 else '   <<<<<< FAIL'}")
    return bad + (err >= 2e-3)


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / 30 * 1e3


def timing():
    extra = [(32, 32, 1280, 640, 640), (32, 32, 640, 640, 640), (32, 32, 1280, 0, 1280), (32, 16, 640, 0, 1280)]
    if "--shallow" in sys.argv:  # (experiments build, CA_WINO_MIN_CIN=320 CA_WINO_MAX_TILES=32768: where does the route stop paying?)
        extra = [(32, 32, 640, 0, 640), (32, 32, 640, 320, 640), (32, 16, 640, 0, 1280), (32, 64, 320, 0, 320), (32, 64, 640, 320, 320), (32, 64, 320, 320, 320)]
    for (images, h, c1, c2, cout) in (extra if "--shallow" in sys.argv else SHAPES + extra):
        d = make(images, h, c1, c2, cout)
        d["post"] = 1.0  # (the resnets' output_scale_factor: the 256 x 320 direct kernel has no post scale)
        K._plan_sink = labels = []
        run(d, True); run(d, False)
        K._plan_sink = None
        row = []
        for _ in range(2):
            row.append(timeit(lambda: run(d, True)))
            row.append(timeit(lambda: run(d, False)))
        print(f"time images={images} {h}x{h} cin={c1 + c2} cout={cout}: {labels[0]} {row[0]:7.1f} {row[2]:7.1f} us   {labels[1]} {row[1]:7.1f} {row[3]:7.1f} us", flush=True)


if __name__ == "__main__":
    rc = 0
    if "--time-only" not in sys.argv:
        rc = check()
    timing()
    sys.exit(1 if rc else 0)
