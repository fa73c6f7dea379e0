"""Persistent streaming kernel (ca_gemm_ps.h): correctness of every epilogue variant against fp32 torch, run-to-run
bit-determinism, and timing at the benchmark shapes.  Needs the experiments library (tuning knobs):

    python -m controlanimate_amd._build --experiments
    CA_HIP_LIB=controlanimate_amd/csrc/libcontrolanimate_hip_exp.so CA_GEMM_PS=1 python tools/ps_check.py --check --time
    CA_HIP_LIB=controlanimate_amd/csrc/libcontrolanimate_hip_exp.so CA_GEMM_PS=0 python tools/ps_check.py --time     # the other kernels

With --check every launch must report the plan label "ps128x320" (else the case does not test what it says).
CA_GEMM_PQ=1 sends every launch the 256 x 320 kernel (ca_gemm_pq.h) can take to it (label "pq256x320"; the others keep theirs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from controlanimate_amd import kernels as K

DEV = "cuda"


def timeit(fn, it=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it


def gemm_reference(kw):
    a = kw["a"].float()
    if kw.get("a2") is not None:
        a = torch.cat([a, kw["a2"].float()], 1)
    dt = kw["a"].dtype
    if kw.get("_ln_ref") is not None:
        st = kw["_ln_ref"]
        a = (a - st[:, :1]) * st[:, 1:]
    y = a @ kw["w"].float().t()
    if kw.get("bias") is not None:
        y = y + kw["bias"]
    if kw.get("rowbias") is not None:
        y = y + kw["rowbias"].repeat_interleave(kw["rows_per_group"], 0)[: y.shape[0]]
    y = (y * kw.get("alpha", 1.0)).to(dt).float()
    if kw.get("residual") is not None:
        y = y + kw["residual"].float()
    y = y * kw.get("post_scale", 1.0)
    if kw.get("act") == 1:
        y = F.silu(y)
    if kw.get("geglu"):
        y = y[:, 0::2] * F.gelu(y[:, 1::2])
    return y


def gemm_cases(dt):
    g = torch.Generator(device="cpu").manual_seed(7)

    def rn(*s, scale=1.0):
        return (torch.randn(*s, generator=g) * scale).to(DEV)

    out = []
    for (m, n, k) in [(8192, 1280, 1280), (8192 - 40, 640, 640), (1000, 640, 320), (33000, 320, 128), (2048, 1280, 2560)]:
        a = rn(m, k).to(dt)
        w = rn(n, k, scale=k ** -0.5).to(dt)
        bias, res = rn(n), rn(m, n).to(dt)
        out.append((f"plain {m}x{n}x{k}", dict(a=a, w=w)))
        out.append((f"bias+res {m}x{n}x{k}", dict(a=a, w=w, bias=bias, residual=res)))
        out.append((f"bias+res+row_sums {m}x{n}x{k}", dict(a=a, w=w, bias=bias, residual=res, row_sums=True)))
        out.append((f"silu post alpha res {m}x{n}x{k}", dict(a=a, w=w, bias=bias, residual=res, act=1, post_scale=0.5, alpha=1.25)))
        rpg = 64 * 3
        rb = rn((m + rpg - 1) // rpg, n)
        out.append((f"rowbias(192) + res {m}x{n}x{k}", dict(a=a, w=w, bias=bias, rowbias=rb, rows_per_group=rpg, residual=res)))
        rb2 = rn((m + 255) // 256, n)
        out.append((f"rowbias(256) + res alpha {m}x{n}x{k}", dict(a=a, w=w, bias=bias, rowbias=rb2, rows_per_group=256, residual=res, alpha=0.75)))
        st = torch.stack([a.float().mean(1), (a.float().var(1, unbiased=False) + 1e-5).rsqrt()], 1).contiguous()
        cs = w.float().sum(1).contiguous()
        out.append((f"LN fold (mean, rstd) {m}x{n}x{k}", dict(a=a, w=w, bias=bias, ln=(st, cs), residual=res, _ln_ref=st)))
        for parts in (1, 2, 4):
            if k % parts:
                continue
            af = a.float().reshape(m, parts, k // parts)
            sums = torch.stack([af.sum(2), (af * af).sum(2)], 2).contiguous()  # [m, parts, 2]
            out.append((f"LN fold, {parts} partial sums {m}x{n}x{k}", dict(a=a, w=w, bias=bias, ln=(K.RowStats(a, 1e-5, (sums, parts)), cs), _ln_ref=st)))
        if n % 640 == 0:
            out.append((f"geglu+bias {m}x{n}x{k}", dict(a=a, w=w, bias=bias, geglu=True)))
            out.append((f"geglu+LN {m}x{n}x{k}", dict(a=a, w=w, bias=bias, geglu=True, ln=(st, cs), _ln_ref=st)))
        if k >= 256:
            k1 = 192 if k > 192 else 64
            out.append((f"two-source {m}x{n}x{k}", dict(a=a[:, :k1].contiguous(), a2=a[:, k1:].contiguous(), w=w, bias=bias)))
        wide = rn(m, k + 2 * n).to(dt)
        out.append((f"strided A/C/res {m}x{n}x{k}", dict(a=wide[:, :k], w=w, residual=wide[:, k:k + n], out=torch.zeros(m, 2 * n, device=DEV, dtype=dt)[:, n:])))
    return out


def conv_cases(dt):
    g = torch.Generator(device="cpu").manual_seed(11)

    def rn(*s, scale=1.0):
        return (torch.randn(*s, generator=g) * scale).to(DEV)

    out = []
    for (img, h, ci, co, stride, ups, c2) in [(4, 32, 640, 1280, 1, 0, 0), (4, 32, 640, 640, 2, 0, 0), (4, 16, 1280, 640, 1, 1, 0), (2, 32, 640, 320, 1, 0, 640), (3, 30, 320, 320, 1, 0, 0), (32, 16, 1280, 1280, 1, 0, 0)]:
        x = rn(img, h, h, ci).to(dt)
        x2 = rn(img, h, h, c2).to(dt) if c2 else None
        w = rn(co, 3, 3, ci + c2, scale=(9 * (ci + c2)) ** -0.5).to(dt)
        ho = (h * (2 if ups else 1) + 2 - 3) // stride + 1
        bias = rn(co)
        rb = rn(img, co)
        res = rn(img, ho, ho, co).to(dt)
        out.append((f"conv {img}x{h}x{h} {ci}+{c2}->{co} s{stride} u{ups}", dict(x=x, x2=x2, w=w, stride=stride, upsample=bool(ups))))
        out.append((f"conv+bias+rowbias+res/post {img}x{h}x{h} {ci}+{c2}->{co} s{stride} u{ups}",
                    dict(x=x, x2=x2, w=w, stride=stride, upsample=bool(ups), bias=bias, rowbias=rb, rows_per_group=ho * ho, residual=res, post_scale=0.7)
                    if (ho * ho) % 64 == 0 else dict(x=x, x2=x2, w=w, stride=stride, upsample=bool(ups), bias=bias, residual=res, post_scale=0.7)))
    return out


def conv_reference(kw):
    x = kw["x"]
    dt = x.dtype
    xin = torch.cat([x, kw["x2"]], 3) if kw.get("x2") is not None else x
    xn = xin.float().permute(0, 3, 1, 2)
    if kw.get("upsample"):
        xn = F.interpolate(xn, scale_factor=2.0, mode="nearest")
    y = F.conv2d(xn, kw["w"].float().permute(0, 3, 1, 2), stride=kw.get("stride", 1), padding=1).permute(0, 2, 3, 1)
    if kw.get("bias") is not None:
        y = y + kw["bias"]
    if kw.get("rowbias") is not None:
        y = y + kw["rowbias"][:, None, None, :]
    y = y.to(dt).float()
    if kw.get("residual") is not None:
        y = y + kw["residual"].float()
    return y * kw.get("post_scale", 1.0)


def check():
    bad = 0
    need_label = os.environ.get("CA_GEMM_PS", "0") not in ("0", "") and os.environ.get("CA_GEMM_PQ", "0") in ("0", "")
    for dt in (torch.float16, torch.bfloat16):
        tol = 2.5e-3 if dt == torch.float16 else 1.5e-2
        for kind, cases, ref_fn, fn in (("gemm", gemm_cases(dt), gemm_reference, K.gemm), ("conv", conv_cases(dt), conv_reference, K.conv3x3)):
            for name, kw in cases:
                call = {k: v for k, v in kw.items() if not k.startswith("_")}
                K._plan_sink = labels = []
                if kind == "conv":
                    x, w = call.pop("x"), call.pop("w")
                    outs = [fn(x, w, **call).clone() for _ in range(3)]
                else:
                    outs = [fn(**call).clone() for _ in range(3)]
                K._plan_sink = None
                ref = ref_fn(kw)
                rel = ((outs[0].float() - ref).norm() / ref.norm()).item()
                same = all(torch.equal(outs[0], o) for o in outs[1:])
                ok = rel < tol and same and bool(torch.isfinite(outs[0].float()).all())
                extra = ""
                if kw.get("row_sums"):
                    o2 = fn(**call)  # (the clones above lost the attribute)
                    rs = K.row_sums_of(o2)
                    if rs is None:
                        extra = " row_sums: NOT PRODUCED"
                        ok = False
                    else:
                        sums, parts = rs
                        of = o2.float().reshape(o2.shape[0], parts, -1)
                        want = torch.stack([of.sum(2), (of * of).sum(2)], 2)
                        e = ((sums - want).norm() / want.norm()).item()
                        extra = f" row_sums rel {e:.1e}"
                        ok = ok and e < 1e-5
                lab = sorted(set(labels))
                if need_label and lab != ["ps128x320"]:
                    extra += f" LABEL {lab}"
                    ok = False
                bad += not ok
                print(f"{str(dt)[6:]:9s} {name:60s} rel {rel:.2e} det={same} {lab}{extra}{'' if ok else '   <<<<<< FAIL'}", flush=True)
    print("FAILED CASES:", bad, flush=True)
    return bad


GEMMS = [(131072, 2560, 320, "geglu"), (131072, 320, 320, "res"), (131072, 320, 320, ""), (131072, 960, 320, ""), (32768, 5120, 640, "geglu"), (32768, 640, 640, "res"),
         (32768, 1920, 640, ""), (8192, 10240, 1280, "geglu"), (8192, 1280, 1280, "res"), (8192, 3840, 1280, ""), (131072, 320, 1280, "res"), (32768, 640, 2560, "res"),
         (8192, 1280, 5120, "res"), (2048, 1280, 1280, "res"), (2048, 10240, 1280, "geglu"), (2048, 3840, 1280, ""), (2048, 1280, 5120, "res")]
CONVS = [(32, 64, 320, 320, 0), (32, 32, 640, 640, 0), (32, 16, 1280, 1280, 0), (32, 8, 1280, 1280, 0), (32, 64, 640, 320, 320), (32, 32, 1280, 640, 640), (32, 16, 2560, 1280, 1280),
         (32, 64, 640, 640, 0), (32, 32, 1280, 1280, 0)]


def times():
    dt = torch.float16
    tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("CA_GEMM") or k.startswith("CA_SPLITK"))
    tot = 0.0
    for (m, n, k, extra) in GEMMS:
        a = torch.randn(m, k, device=DEV).to(dt)
        w = (torch.randn(n, k, device=DEV) * k ** -0.5).to(dt)
        kw = dict(bias=torch.randn(n, device=DEV))
        if extra == "res":
            kw["residual"] = torch.randn(m, n, device=DEV).to(dt)
        if extra == "geglu":
            kw["geglu"] = True
        K._plan_sink = lab = []
        K.gemm(a, w, **kw)
        K._plan_sink = None
        ms = timeit(lambda: K.gemm(a, w, **kw))
        tot += ms
        print(f"gemm {m}x{n}x{k} {extra:5s}: {ms * 1e3:8.1f} us {2.0 * m * n * k / ms / 1e9:7.1f} TF  {lab[0]:18s} [{tag}]", flush=True)
    for (img, h, ci, co, c2) in CONVS:
        x = torch.randn(img, h, h, ci - c2, device=DEV).to(dt)
        x2 = torch.randn(img, h, h, c2, device=DEV).to(dt) if c2 else None
        w = (torch.randn(co, 3, 3, ci, device=DEV) * (9 * ci) ** -0.5).to(dt)
        kw = dict(bias=torch.randn(co, device=DEV), residual=torch.randn(img, h, h, co, device=DEV).to(dt))
        K._plan_sink = lab = []
        K.conv3x3(x, w, x2=x2, **kw)
        K._plan_sink = None
        ms = timeit(lambda: K.conv3x3(x, w, x2=x2, **kw))
        tot += ms
        print(f"conv {img}x{h}x{h} {ci}->{co}: {ms * 1e3:8.1f} us {2.0 * img * h * h * co * 9 * ci / ms / 1e9:7.1f} TF  {lab[0]:18s} [{tag}]", flush=True)
    print(f"sum of the listed launches: {tot:.3f} ms [{tag}]", flush=True)


if __name__ == "__main__":
    rc = 0
    if "--check" in sys.argv:
        rc = check()
    if "--time" in sys.argv:
        times()
    sys.exit(1 if rc else 0)
