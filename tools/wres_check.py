"""Weight-resident K=320 GEMM (ca_gemm_wres.h): correctness against fp32 torch and against the tiled kernels
(CA_GEMM_WRES=0 in a child process is not needed: the env is read once per process, so this script runs the SAME cases
in two processes and compares saved outputs), determinism, and timing.
    python tools/wres_check.py            # both passes + comparison + timing
"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def cases(torch, dev, dt, small=False):
    g = torch.Generator(device="cpu").manual_seed(7)
    def rn(*s, scale=1.0):
        return (torch.randn(*s, generator=g) * scale).to(dev)
    out = []
    for (m, n) in ([(32768 + 40, 320), (20000 - 24, 960), (16384 + 8, 2560)] if small else [(131072, 320), (40000 - 24, 960), (16384 + 8, 2560), (65536, 1280)]):
        k = 320
        a = rn(m, k).to(dt); w = rn(n, k, scale=k ** -0.5).to(dt)
        bias = rn(n); res = rn(m, n).to(dt)
        out.append((f"plain {m}x{n}", dict(a=a, w=w)))
        out.append((f"bias+res {m}x{n}", dict(a=a, w=w, bias=bias, residual=res)))
        if n == 2560:
            out.append((f"geglu+bias {m}x{n}", dict(a=a, w=w, bias=bias, geglu=True)))
        if n == 960:
            rpg = 4096 if m % 4096 == 0 else 32 * 39
            rb = rn((m + rpg - 1) // rpg, n)
            out.append((f"rowbias {m}x{n}", dict(a=a, w=w, rowbias=rb, rows_per_group=rpg)))
            a1 = a[:, :192].contiguous(); a2 = a[:, 192:].contiguous()
            out.append((f"two-source silu post {m}x{n}", dict(a=a1, a2=a2, w=w, bias=bias, act=1, post_scale=0.5, alpha=1.25)))
        if n == 320:
            st = torch.stack([a.float().mean(1), (a.float().var(1, unbiased=False) + 1e-5).rsqrt()], 1).contiguous()
            cs = w.float().sum(1).contiguous()
            out.append((f"LN fold {m}x{n}", dict(a=a, w=w, bias=bias, ln=(st, cs), residual=res)))
            out.append((f"LN fold no-res {m}x{n}", dict(a=a, w=w, bias=bias, ln=(st, cs))))
            out.append((f"LN fold, statistics inside the GEMM {m}x{n}", dict(a=a, w=w, bias=bias, ln=("inline", cs), residual=res, _ln_ref=st)))
            out.append((f"silu post alpha {m}x{n}", dict(a=a, w=w, bias=bias, act=1, post_scale=0.5, alpha=1.25)))
            wide = rn(m, 1280).to(dt)
            out.append((f"strided A/C {m}x{n}", dict(a=wide[:, 320:640], w=w, residual=wide[:, 640:960], out=torch.zeros(m, 640, device=dev, dtype=dt)[:, 320:])))
    return out


def reference(torch, F, kw):
    a = kw["a"].float()
    if kw.get("a2") is not None:
        a = torch.cat([a, kw["a2"].float()], 1)
    dt = kw["a"].dtype
    if kw.get("ln") is not None:
        st, cs = kw["ln"]
        if isinstance(st, str):
            st = kw["_ln_ref"]
        a = (a - st[:, :1]) * st[:, 1:]
    y = a @ kw["w"].float().t()
    if kw.get("bias") is not None: y = y + kw["bias"]
    if kw.get("rowbias") is not None:
        y = y + kw["rowbias"].repeat_interleave(kw["rows_per_group"], 0)[: y.shape[0]]
    y = (y * kw.get("alpha", 1.0)).to(dt).float()
    if kw.get("residual") is not None: y = y + kw["residual"].float()
    y = y * kw.get("post_scale", 1.0)
    if kw.get("act") == 1: y = F.silu(y)
    if kw.get("geglu"): y = y[:, 0::2] * F.gelu(y[:, 1::2])
    return y


def call(K, kw):
    """K.gemm(**kw); ln=("inline", colsum) asks for the statistics inside the GEMM (K.RowStats)."""
    kw = {k: v for k, v in kw.items() if not k.startswith("_")}
    if kw.get("ln") is not None and isinstance(kw["ln"][0], str):
        kw["ln"] = (K.RowStats(kw["a"], 1e-5), kw["ln"][1])
    return K.gemm(**kw)


def run(tag):
    import torch, torch.nn.functional as F
    from controlanimate_amd import kernels as K
    dev = "cuda"
    bad = 0
    saved = {}
    for dt in (torch.float16, torch.bfloat16):
        for name, kw in cases(torch, dev, dt):
            outs = [call(K, kw).clone() for _ in range(3)]
            ref = reference(torch, F, kw)
            rel = ((outs[0].float() - ref).norm() / ref.norm()).item()
            same = all(torch.equal(outs[0], o) for o in outs[1:])
            tol = 2e-3 if dt == torch.float16 else 1.2e-2
            flag = "" if rel < tol and same and torch.isfinite(outs[0].float()).all() else "   <<<<<< FAIL"
            bad += bool(flag)
            print(f"[{tag}] {str(dt)[6:]:9s} {name:36s} rel {rel:.2e} deterministic={same}{flag}", flush=True)
            saved[f"{dt}|{name}"] = outs[0].cpu()
    torch.save(saved, f"/tmp/wres_{tag}.pt")
    # timing
    def timeit(fn, it=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(it): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / it * 1e3
    dt = torch.float16
    for (m, n, extra) in [(131072, 320, "res"), (131072, 320, ""), (131072, 960, ""), (131072, 2560, "geglu"), (131072, 1280, ""), (32768, 320, "res")]:
        a = torch.randn(m, 320, device=dev).to(dt); w = (torch.randn(n, 320, device=dev) * 320 ** -0.5).to(dt)
        bias = torch.randn(n, device=dev); res = torch.randn(m, n, device=dev).to(dt)
        kw = dict(bias=bias)
        if extra == "res": kw["residual"] = res
        if extra == "geglu": kw["geglu"] = True
        us = timeit(lambda: K.gemm(a, w, **kw))
        print(f"[{tag}] time {m}x{n}x320 {extra:6s} {us:8.1f} us  {2 * m * n * 320 / us * 1e-6:7.1f} TF", flush=True)
    return bad


if __name__ == "__main__":
    if len(sys.argv) > 1:
        sys.exit(1 if run(sys.argv[1]) else 0)
    rc = 0
    for tag, env in (("wres", "1"), ("tiled", "0")):
        e = dict(os.environ, CA_GEMM_WRES=env)  # ("1": weight-resident / weights-in-registers kernels; "0": tiled kernels)
        rc |= subprocess.call([sys.executable, os.path.abspath(__file__), tag], env=e)
    import torch
    a, b = torch.load("/tmp/wres_wres.pt"), torch.load("/tmp/wres_tiled.pt")
    for k in a:
        d = (a[k].float() - b[k].float()).abs().max().item()
        neq = (a[k] != b[k]).float().mean().item()
        print(f"wres vs tiled {k:60s} max|d| {d:.3e}  differing {100 * neq:.3f}%")
    sys.exit(rc)
