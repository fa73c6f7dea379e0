#!/usr/bin/env python3
"""Benchmark of the denoising-loop hot path (BASELINE.json metric: frames/sec + sec/denoise-step).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is ONE iteration of the reference's denoising loop
(animatediff/pipelines/controlanimation_pipeline.py:790-855) on one 16-frame window:
ControlNet residuals + UNet3D eps (CFG batch 2) + fused CFG combine / scheduler update.
Workload at N=1 (BASELINE.json configs[1]): SD1.5 + mm_sd_v15_v2, 16 frames, 512x512 (latent 64x64),
20 diffusers-LCM steps, guidance 1.1 (CFG on), 1 ControlNet (canny-shaped hints), synthetic seeded
weights/inputs (no checkpoints exist offline).  Every rank runs its own window (windows are the
sharding unit, weak scaling); weights are broadcast once over RCCL before timing.

frames/sec = n_gpus * frames_per_window / (steps_per_window * sec_per_step), steps_per_window = 20.
Default K = one whole window (the timed region begins at a window start and so contains the per-window work once per
20 steps, the product's own ratio); W = 3 warm-up steps are the tail of the window before it.
Prints one JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_TFLOPS = 2500.0  # dense bf16/fp16 MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBPS = 8000.0     # HBM3E, MI355X_MICROARCH.md

# BASELINE.json configs.  tflop = algorithmic work per denoise step (2*MAC of conv, linear,
# QK^T, PV): UNet3D + n_controlnets x ControlNet at the batch the reference feeds it (guess mode / native LCM: b = 1).
CONFIGS = {
    1: dict(name="config 1: SD1.5 + mm_sd_v15 (v1), 8 frames 256x256, 4 DDIM steps, CFG 7.5, no ControlNet", version="v1", frames=8,
            height=256, width=256, steps=4, scheduler="DDIMScheduler", guidance=7.5, controlnets=0, guess_mode=False, ip=False, overlap=0,
            unet_tflop=4.09, cn_tflop=0.0),
    2: dict(name="config 2: SD1.5 + mm_sd_v15_v2, 16 frames 512x512, 20 LCM steps, CFG g=1.1 (batch 2), 1 ControlNet", version="v2", frames=16,
            height=512, width=512, steps=20, scheduler="LCMScheduler", guidance=1.1, controlnets=1, guess_mode=False, ip=False, overlap=0,
            unet_tflop=35.53, cn_tflop=9.07),
    3: dict(name="config 3: SampleConfig.yaml-equivalent, 4 ControlNets + 8-frame latent overlap, 16 frames 512x512, Euler 30 steps, CFG 7.5",
            version="v2", frames=16, height=512, width=512, steps=30, scheduler="EulerDiscreteScheduler", guidance=7.5, controlnets=4,
            guess_mode=False, ip=False, overlap=8, unet_tflop=35.53, cn_tflop=9.07),
    4: dict(name="config 4: IP-Adapter (81 tokens) + LCM-LoRA + 2 ControlNets (guess mode), 16 frames 512x768, 20 LCM steps, CFG g=1.35",
            version="v2", frames=16, height=512, width=768, steps=20, scheduler="LCMScheduler", guidance=1.35, controlnets=2, guess_mode=True,
            ip=True, overlap=0, unet_tflop=56.20, cn_tflop=7.38),
    5: dict(name="config 5: mm_sd_v15_v2, 32 frames 768x768 with 4-frame overlap blending, 20 LCM steps, CFG g=1.1, no ControlNet",
            version="v2", frames=32, height=768, width=768, steps=20, scheduler="LCMScheduler", guidance=1.1, controlnets=0, guess_mode=False,
            ip=False, overlap=4, unet_tflop=181.91, cn_tflop=0.0),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0,
                    help="timed steps K; default (0) = the steps of ONE window of the config (20 for config 2): the timed region begins at "
                         "a window start, so a whole window holds the per-window work exactly once, as the product's loop does")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json config (2 = the headline workload)")
    ap.add_argument("--frames", type=int, default=None, help="override the config's frames per window")
    ap.add_argument("--size", type=int, default=None, help="override: frame height = width in pixels")
    ap.add_argument("--controlnets", type=int, default=None, help="override the config's number of ControlNets")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-config2", action="store_true", help=argparse.SUPPRESS)  # (old spelling: now the default)
    ap.add_argument("--no-cpu-baseline-config2", action="store_true",
                    help="skip the full config-2 denoise step of the fp32 oracle on the host cores (44.6 TFLOP, ControlNet + UNet3D: ~100 s "
                         "of CPU work and ~40 GB of host memory; measured once per host and cached in the temp directory; skipped by "
                         "itself when less than 48 GB of host memory are available)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short passes of BASELINE configs 1, 3, 4, 5 and bf16 config 2 that the default single-GPU config-2 run "
                         "appends as `other_configs` (outside `value`; ~2 min)")
    ap.add_argument("--shapes", action="store_true", help="also print the per-shape GEMM/conv table (stderr)")
    ap.add_argument("--no-vae", action="store_true", help="skip the VAE encode/decode timing (reported beside the metric)")
    ap.add_argument("--no-overlap", action="store_true", help="run the ControlNet stack on the main stream (no 2nd-stream overlap)")
    ap.add_argument("--no-fuse-adds", action="store_true", help="13 separate ControlNet residual adds instead of the zero convolutions' epilogues (A/B)")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch every kernel eagerly.  Default: the ControlNet + UNet part of a step is captured once as a "
                         "hipGraph (both streams) and replayed -- same kernels, bit-identical results "
                         "(tests/test_graph_gpu.py), ~1 ms instead of ~50 ms of host time per step, so the loop stays "
                         "GPU-bound when N ranks share one host; falls back to eager if the capture fails")
    ap.add_argument("--graph", action="store_true", help=argparse.SUPPRESS)  # (old spelling of the default)
    ap.add_argument("--steps-in-flight", type=int, default=None,
                    help="ControlAnimationPipeline.steps_in_flight: how many denoise steps the host may enqueue ahead of the device "
                         "before it sleeps on a blocking HIP event (default: the pipeline's own, 2; 0 = unpaced, the round-4 behaviour "
                         "-- the host then spins in the runtime for most of every step)")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="multi-rank plumbing rehearsal WITHOUT the hot path (runs on a CPU box over gloo): rendezvous, "
                         "weight-arena broadcast, barriers, max-over-ranks timing, rank-0 JSON with `dry_run: true` and "
                         "`value: null`.  Never a measurement; used by tests/test_bench_launch.py")
    ap.add_argument("--chains", type=int, default=2,
                    help="after the headline (single GPU, outside `value`): N independent windows in flight on the one GPU (controlanimate_amd/chains.py), "
                         "reported as `two_chains`; 1 skips it")
    ap.add_argument("--window-graph", action="store_true",
                    help="ControlAnimationPipeline.window_graph: a whole window as ONE captured hipGraph, one replay per window, no per-step events (A/B; "
                         "slower on ROCm 7.2 -- the launch cost of a graph grows with the square of its node count).  Default: one replay + three "
                         "eager launches per STEP, a HIP event after every step")
    ap.add_argument("--pace-wait", default=None, choices=["sleep", "event"],
                    help="how the paced loop waits: hipEventQuery polls between naps (default) or a blocking hipEventSynchronize (A/B)")
    args = ap.parse_args()
    if args.steps <= 0:
        args.steps = int(CONFIGS[args.config]["steps"])
    return args


def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a torchrun environment: this process becomes a GPU-free parent
    that starts N fresh rank processes (one per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment, 127.0.0.1 rendezvous), relays rank 0's JSON line and returns non-zero if any rank fails.
    Nothing here touches HIP: a process that has initialised the GPU must never be re-executed, and
    children are started with subprocess (fresh interpreters), not fork."""
    import subprocess
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # watch EVERY rank: if one dies before the rendezvous the others would sit in it until the process-group timeout
    import threading
    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    while True:
        codes = [pr.poll() for pr in procs]
        failed = [c for c in codes if c not in (None, 0)]
        if failed:
            rc = failed[0]
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.2)
    if rc:
        time.sleep(1.0)
        for r, pr in enumerate(procs):
            if pr.poll() is None:
                pr.kill()  # exact PID of a child this parent started
        print(f"[bench] a rank exited with code {rc}; the remaining ranks were stopped", file=sys.stderr)
    for pr in procs:
        pr.wait()
    reader.join(timeout=5)
    sys.stdout.write("".join(b for b in buf if b))
    sys.stdout.flush()
    return rc


def randomize_zero_init_(model, std=0.02, seed=0):
    """Real checkpoints are non-zero where the architecture zero-initialises (motion proj_out,
    ControlNet zero-convs): all-zero operands would also let the chip clock higher (DVFS)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    for p in model.parameters():
        if p.numel() and float(p.detach().abs().max()) == 0.0 and p.dim() > 1:
            p.data.copy_((torch.randn(p.shape, generator=g) * std).to(p.device))


def rocprof_kernel_name(family: str, dtype: str) -> str:
    """The demangled kernel name rocprofv3 reports for a family label (profiles/*kernel_stats.csv).  Family labels are
    `<op>_<plan>` with the plan label the library itself reports for a launch (ca_gemm_plan_name / ca_conv3x3_plan_name)."""
    op, tile = family.split("_", 1)
    if tile.startswith("reg_") or tile == "?":
        return ""
    dt = 1 if dtype == "fp16" else 0
    mode = 1 if op == "conv3x3" else 0
    if tile == "wres160":
        return f"k_gemm_wres<{dt}>"
    if tile == "ar128x64":  # (every epilogue variant: k_gemm_ar<dt, EPI>)
        return f"k_gemm_ar<{dt}, "
    if tile in ("fused128", "out128") and op == "ff":  # "ff_fused128" / "ff_out128": the feed-forward in one launch (+ proj_out, round 5)
        return f"k_ff_fused<{dt}, {'true' if tile == 'out128' else 'false'}>"
    if tile.startswith("wino_"):  # Winograd route: three kernels per call (k_wino_in, k_gemm_pq<dt, 0, 0>, k_wino_out) -- no single name
        return ""
    if tile.startswith("pp128x320"):
        return f"k_gemm_pp2<{dt}, {mode}"
    if tile == "ps128x320":
        return f"k_gemm_ps<{dt}, {mode}"
    if tile.startswith("pq256x320"):  # (every epilogue variant of the family: k_gemm_pq<dt, mode, EPI>)
        return f"k_gemm_pq<{dt}, {mode}"
    base = tile.split("_")[0]
    if base.count("x") != 1 or not base.replace("x", "").isdigit():
        return ""  # (a label this table does not know: no rocprof name, never a crash of the bench line)
    bm, bn = base.split("x")
    waves = "4, 1" if base == "128x64" else "2, 2"
    nbuf = 2 if tile.endswith("_db") else 3 if tile.endswith("_r3") else 4 if tile.endswith("_r4") else 1
    return f"k_gemm_dma<{dt}, {bm}, {bn}, {waves}, {mode}, {nbuf}, "


def kernel_display_name(family: str) -> str:
    tile = family.split("_", 1)[1]
    if tile == "wres160":
        return f"k_gemm_wres<{family}>"
    if tile == "ar128x64":
        return f"k_gemm_ar<{family}>"
    if tile in ("fused128", "out128") and family.startswith("ff_"):
        return f"k_ff_fused<{family}>"
    if tile.startswith("wino_"):
        return f"k_wino_in + k_gemm_pq + k_wino_out<{family}>"
    if tile.startswith("pp128x320"):
        return f"k_gemm_pp2<{family}>"
    if tile == "ps128x320":
        return f"k_gemm_ps<{family}>"
    if tile.startswith("pq256x320"):
        return f"k_gemm_pq<{family}>"
    return f"k_gemm_dma<{family}>"


PMC_SUMMARIES = ("round6_pmc_traffic.json", "round5_pmc_traffic.json", "round4_pmc_traffic.json", "round3_pmc_traffic.json", "round2_pmc_traffic.json", "round1_pmc_traffic.json")


def pmc_traffic(kernel_prefix: str, workload_key: str, dtype: str):
    """HBM-side bytes per launch of a kernel from a committed PMC summary (profiles/roundN_pmc_traffic.json: rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `bench.py`, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
    for gfx950).  Counters cannot be collected by this process itself, so the figure is attached ONLY when the summary
    was captured on the same workload and dtype (the summary records both); otherwise (None, None).
    Returns (bytes_per_launch, source file)."""
    base = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    if not kernel_prefix:
        return None, None
    for fn in PMC_SUMMARIES:
        path = os.path.join(base, fn)
        if not os.path.exists(path):
            continue
        with open(path) as fh:
            doc = json.load(fh)
        meta = doc.get("workload", {"key": "config2", "dtype": "fp16"})  # (round 1 summary: config 2, fp16)
        if meta.get("key") != workload_key or meta.get("dtype") != dtype:
            continue
        # (a family can be several template instantiations -- the epilogue variants of k_gemm_pq: launch-weighted mean)
        rows = [row for name, row in doc["kernels"].items() if name.startswith(kernel_prefix)]
        n = sum(r.get("launches", 1) for r in rows)
        if rows and n > 0:
            return int(sum(r["hbm_bytes_per_launch"] * r.get("launches", 1) for r in rows) / n), fn
    return None, None


class KernelTimer:
    """HIP-event timing of every ca_gemm / ca_conv3x3 launch (events are recorded on the stream the
    kernels are launched on: torch's current stream) + the algorithmic FLOPs of each launch."""

    def __init__(self):
        self.records = []  # (variant, flops, start_event, end_event, shape)
        self.other = []
        self.bytes = {}  # id(start event) -> algorithmic bytes of the launch
        self.enabled = False

    def install(self):
        from controlanimate_amd import kernels as K
        gemm0, conv0 = K.gemm, K.conv3x3
        timer = self

        def gemm(a, w, **kw):
            if not timer.enabled:
                return gemm0(a, w, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            K._plan_sink = sink = []
            s.record()
            out = gemm0(a, w, **kw)
            e.record()
            K._plan_sink = None
            m, n, k = a.shape[0], w.shape[0], w.shape[1]
            # algorithmic bytes: every operand once (A, W, C, residual), 2 B per element
            nb = 2.0 * (m * k + n * k + m * (n // 2 if kw.get("geglu") else n) + (m * n if kw.get("residual") is not None else 0))
            timer.bytes[id(s)] = nb
            timer.records.append((f"gemm_{sink[-1] if sink else '?'}", 2.0 * m * n * k, s, e, (m, n, k)))
            return out

        def conv3x3(x, w, **kw):
            if not timer.enabled:
                return conv0(x, w, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            K._plan_sink = sink = []
            s.record()
            out = conv0(x, w, **kw)
            e.record()
            K._plan_sink = None
            n = w.shape[0]
            mrows = out.shape[0] * out.shape[1] * out.shape[2]
            timer.bytes[id(s)] = 2.0 * (x.numel() + (kw["x2"].numel() if kw.get("x2") is not None else 0) + w.numel() + out.numel())
            timer.records.append((f"conv3x3_{sink[-1] if sink else '?'}", 2.0 * mrows * n * 9 * w.shape[3], s, e,
                                  (mrows, n, 9 * w.shape[3])))
            return out

        ff0 = K.ff_fused

        def ff_fused(x, w1_frag, bias1, colsum1, w2_frag, bias2, ln_eps, **kw):
            if not timer.enabled:
                return ff0(x, w1_frag, bias1, colsum1, w2_frag, bias2, ln_eps, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = ff0(x, w1_frag, bias1, colsum1, w2_frag, bias2, ln_eps, **kw)
            e.record()
            if out is not None:  # (None: the library does not take the arguments, the caller runs the two GEMMs -- timed there)
                m, c = x.shape
                inner = w2_frag.shape[1]
                # x, y (+ residual) once, both weight matrices once; the [m, inner] intermediate never leaves the CU
                with_out = kw.get("w_out_frag") is not None  # round 5: + the transformer's proj_out (+ its residual) in the same launch
                rows_io = (3 if kw.get("residual") is not None else 2) + (1 if with_out and kw.get("residual_out") is not None else 0)
                timer.bytes[id(s)] = 2.0 * (m * c * rows_io + 3 * inner * c + (c * c if with_out else 0))
                timer.records.append(("ff_out128" if with_out else "ff_fused128", 2.0 * m * c * 2 * inner + 2.0 * m * inner * c + (2.0 * m * c * c if with_out else 0.0),
                                      s, e, (m, 3 * inner + (c if with_out else 0), c)))
            return out

        K.gemm, K.conv3x3, K.ff_fused = gemm, conv3x3, ff_fused

        def wrap_other(name, shape_of):
            fn0 = getattr(K, name)

            def fn(*a, **kw):
                if not timer.enabled:
                    return fn0(*a, **kw)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                out = fn0(*a, **kw)
                e.record()
                timer.other.append((name, shape_of(*a, **kw), s, e))
                return out
            setattr(K, name, fn)

        wrap_other("attention_raw", lambda q, k, v, o, **kw: (kw["batches"], kw["heads"], kw["head_dim"], kw["nq"], kw["nk"]))
        wrap_other("group_norm", lambda x, g, b, **kw: tuple(x.shape) + ((kw["x2"].shape[3],) if kw.get("x2") is not None else (0,)))
        wrap_other("layer_norm", lambda x, g, b, **kw: tuple(x.shape))
        wrap_other("row_stats", lambda x, *a, **kw: tuple(x.shape))
        wrap_other("add_bcast", lambda a, b, out=None: (a.numel(),))

    def other_summary(self, top=30):
        agg = {}
        for name, shape, s, e in self.other:
            d = agg.setdefault((name,) + tuple(shape), dict(n=0, ms=0.0))
            d["n"] += 1
            d["ms"] += s.elapsed_time(e)
        rows = sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:top]
        return [dict(op=k[0], shape=k[1:], launches=v["n"], ms=round(v["ms"], 3), avg_us=round(1e3 * v["ms"] / v["n"], 1)) for k, v in rows]

    def by_shape(self, top=40):
        agg = {}
        for name, flops, s, e, shape in self.records:
            d = agg.setdefault((name,) + shape, dict(n=0, flops=0.0, ms=0.0, each=[]))
            d["n"] += 1
            d["flops"] += flops
            d["each"].append(s.elapsed_time(e))
        for d in agg.values():  # the median launch x the launch count: the instrumented pass runs eagerly, and the first launch of a shape can
            d["each"].sort()    # carry a one-off allocation of its scratch (20 ms for the 1 GB of a Winograd convolution's V and M)
            d["med"] = d["each"][len(d["each"]) // 2]
            d["ms"] = d["med"] * d["n"]
        rows = sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:top]
        return [dict(op=k[0], m=k[1], n=k[2], k=k[3], launches=v["n"], ms=round(v["ms"], 3), median_us=round(1e3 * v["med"], 1),
                     tflops=round(v["flops"] / v["n"] / (v["med"] * 1e-3) / 1e12, 1)) for k, v in rows]

    def summary(self):
        agg = {}
        for name, flops, s, e, _shape in self.records:
            d = agg.setdefault(name, dict(launches=0, flops=0.0, ms=0.0, bytes=0.0))
            d["launches"] += 1
            d["flops"] += flops
            d["bytes"] += self.bytes.get(id(s), 0.0)
            d["ms"] += s.elapsed_time(e)
        for d in agg.values():
            d["avg_us"] = 1e3 * d["ms"] / d["launches"]
            d["tflops"] = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
        return agg


def build_models(wl, device, dtype, rank: int = 0):
    """rank > 0 (multi-rank runs): DIFFERENT random weights, so that the run only works -- and the arena checksums only agree --
    because rank 0's arenas really arrive over the broadcast (in a single-process run nothing changes)."""
    from controlanimate_amd.configs import controlnet_config, unet_config
    from controlanimate_amd.controlnet import ControlNetModel
    from controlanimate_amd.unet import UNet3DConditionModel
    torch.manual_seed(1000 * rank)
    with torch.device(device):
        unet = UNet3DConditionModel.from_config(unet_config(wl["version"]))
        nets = [ControlNetModel.from_config(controlnet_config()) for _ in range(wl["controlnets"])]
    randomize_zero_init_(unet, seed=1)
    for i, n in enumerate(nets):
        randomize_zero_init_(n, seed=2 + i)
    if wl["ip"]:  # IP-Adapter processors at the 16 cross-attention sites + CN processors on the ControlNets (modules/ip_adapter.py:95-134)
        from types import SimpleNamespace
        from controlanimate_amd.ip_adapter import IPAdapter
        ip = IPAdapter(SimpleNamespace(unet=unet), None, None, device, num_tokens=4)
        g = torch.Generator().manual_seed(77)
        for proc in unet.attn_processors.values():
            if hasattr(proc, "to_k_ip"):
                for lin in (proc.to_k_ip, proc.to_v_ip):
                    lin.weight.data.copy_((torch.randn(lin.weight.shape, generator=g) * lin.weight.shape[1] ** -0.5).to(device))
        ip.set_scale(0.4)
        if nets:
            ip.set_ip_adapter_4controlanimate(SimpleNamespace(controlnet=SimpleNamespace(nets=nets)))
    unet.prepare(device, dtype)
    for n in nets:
        n.prepare(device, dtype)
    return unet, nets


def usable_cores() -> int:
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 32))  # >32 threads only adds synchronisation overhead at these op sizes


def cpu_baseline_config2(threads: int):
    """ONE full config-2 denoise step ("config 1 in full and one step of config 2") with the fp32 oracle:
    ControlNet (B = 32) + UNet3D v2 (b = 2, f = 16, 64x64 latents) on the host cores.  ~44.6 TFLOP."""
    from oracle.controlnet import ControlNetConfig, init_controlnet_weights, multi_controlnet_residuals
    from oracle.unet3d import UNet3DConfig, init_unet3d_weights, unet3d_forward
    torch.set_num_threads(threads)
    ucfg, ccfg = UNet3DConfig.v2(), ControlNetConfig()
    uw, cw = init_unet3d_weights(ucfg, seed=0), init_controlnet_weights(ccfg, seed=1)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 4, 16, 64, 64, generator=g)
    ehs = torch.randn(2, 77, 768, generator=g) * 0.5
    hints = torch.rand(32, 3, 512, 512, generator=g)
    with torch.no_grad():
        t0 = time.time()
        down, mid = multi_controlnet_residuals([cw], ccfg, x, 999, ehs, 16, [hints], [1.0], guess_mode=False)
        unet3d_forward(uw, ucfg, x, 999, ehs, down, mid)
        dt = time.time() - t0
    return {"sec_per_step": dt, "tflops": 44.6 / dt, "frames_per_sec": 16.0 / (20 * dt), "cores": threads,
            "sample": "1 oracle denoise step of BASELINE config 2 (ControlNet B=32 + UNet3D b=2 f=16 64x64 latents, fp32)"}


def cpu_baseline_config2_cached(threads: int):
    """cpu_baseline_config2 once per host and core count (the file lives in the temp directory), None when the host has
    too little free memory for the fp32 oracle at this size."""
    import socket
    import tempfile
    path = os.path.join(tempfile.gettempdir(), f"controlanimate_amd_cpu_config2_{socket.gethostname()}_{threads}.json")
    try:
        with open(path) as fh:
            d = json.load(fh)
        d["cached"] = True
        return d
    except (OSError, ValueError):
        pass
    try:
        avail_kb = next(int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable:"))
    except (OSError, StopIteration, ValueError):
        avail_kb = 0
    if avail_kb < 48 * 1024 * 1024:
        return {"skipped": "less than 48 GB of host memory available (%d MB)" % (avail_kb // 1024)}
    d = cpu_baseline_config2(threads)
    try:
        with open(path, "w") as fh:
            json.dump(d, fh)
    except OSError:
        pass
    d["cached"] = False
    return d


def cpu_baseline(threads: int):
    """The oracle (fp32 restatement of the reference, oracle/) timed on the host cores on a bounded
    sample: ONE UNet3D forward at the config-1 shape (2,4,8,32,32), L=77, mm v1 -- the same probe as
    BASELINE.md section 2 (reference's own code: 8.35 s on 8 cores)."""
    from oracle.unet3d import UNet3DConfig, init_unet3d_weights, unet3d_forward
    torch.set_num_threads(threads)
    cfg = UNet3DConfig.v1()
    w = init_unet3d_weights(cfg, seed=0)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 4, 8, 32, 32, generator=g)
    ehs = torch.randn(2, 77, 768, generator=g) * 0.5
    with torch.no_grad():
        t0 = time.time()
        unet3d_forward(w, cfg, x, 500, ehs)
        dt = time.time() - t0
    flops = 4.09e12
    return {"value": 8.0 / (4 * dt), "unit": "frames/s (config-1 shape: 8 frames 256x256, 4 DDIM steps, UNet3D only)",
            "cores": threads, "kind": "port", "sec_per_step": dt, "tflops": flops / dt / 1e12,
            "sample": "1 oracle UNet3D forward (2,4,8,32,32) fp32, mm v1, = 1 denoise step of BASELINE config 1 (4.09 TFLOP); "
                      "config-2 step is 44.6 TFLOP => FLOP-scaled estimate %.1f s/step" % (dt * 44.6 / 4.09)}


def time_vae(wl, device, dtype):
    """ms per window for encoding / decoding all frames with the HIP AutoencoderKL."""
    from controlanimate_amd.vae import AutoencoderKL
    torch.manual_seed(0)
    vae = AutoencoderKL.from_config().to(device).prepare(device, dtype)
    g = torch.Generator().manual_seed(4321)
    imgs = (torch.rand(wl["frames"], 3, wl["height"], wl["width"], generator=g) * 2 - 1).to(device)
    lat = torch.randn(wl["frames"], 4, wl["height"] // 8, wl["width"] // 8, generator=g).to(device)
    out = {}
    for name, fn in (("encode_ms_per_window", lambda: vae.encode_moments(imgs)), ("decode_ms_per_window", lambda: vae.decode(lat))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        out[name] = round((time.perf_counter() - t0) / 3 * 1e3, 2)
    del vae
    torch.cuda.empty_cache()
    return out


def plumbing_only(args):
    """The multi-rank flow of main() with the hot path replaced by a sleep: what a CPU box can rehearse."""
    from controlanimate_amd import window_shard as WS
    rank, world, _ = WS.init_distributed()
    g = torch.Generator().manual_seed(11)
    arenas = [torch.randint(0, 255, (1 << 20,), dtype=torch.uint8, generator=g) if rank == 0 else torch.zeros(1 << 20, dtype=torch.uint8)
              for _ in range(1 + CONFIGS[args.config]['controlnets'])]
    ref = int(arenas[0][:4096].to(torch.int64).sum()) if rank == 0 else None
    bytes_bcast = WS.broadcast_weights(arenas)
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001)
    if world > 1:
        torch.distributed.barrier()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    chk = torch.tensor([int(arenas[0][:4096].to(torch.int64).sum())], dtype=torch.int64)
    if world > 1:
        torch.distributed.all_reduce(elapsed, op=torch.distributed.ReduceOp.MAX)
        lo = chk.clone()
        torch.distributed.all_reduce(chk, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        assert int(chk) == int(lo), "ranks disagree on the broadcast arena"
    if rank == 0:
        assert ref == int(chk)
        print(json.dumps({"metric": "frames_per_sec", "value": None, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": None, "dry_run": True, "scaling": "weak",
                          "config": {"workload": "plumbing rehearsal (no hot path)", "weight_broadcast_bytes": bytes_bcast,
                                     "parallelism": f"window-shard x{world}" if world > 1 else "single GPU"},
                          "max_rank_elapsed_s": float(elapsed)}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()



def matrix_rate_vs_operand_data(dtype):
    """What the matrix pipes of THIS box sustain on real data (reported beside `roofline`, never part of `value`).

    MI355X clocks and issues to its power budget: the same GEMM binary runs ~1.5x faster on zero-filled operands than on
    N(0,1) data (DESIGN.md section 3, round-3 findings).  The guide's 2.5 PFLOP/s dense peak is therefore not reachable on the
    activations this workload multiplies; the rates below say what is: the vendor library's best case (`torch.matmul`, hipBLASLt,
    8192^3 -- a measurement reference only, the product never calls it) and this package's 256 x 320 kernel on one of the
    step's own shapes, each on N(0,1) and on zero-filled operands.
    """
    from controlanimate_amd import kernels as K

    def rate(fn, flop, it=20):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return round(flop * it / (e0.elapsed_time(e1) * 1e-3) / 1e12, 1)

    res = {"unit": "TFLOP/s", "dtype": str(dtype).replace("torch.", "")}
    g = torch.Generator(device="cuda").manual_seed(1)
    for name, (m, n, k), ours in (("vendor_gemm_8192x8192x8192", (8192, 8192, 8192), False), ("pq256x320_32768x640x2560", (32768, 640, 2560), True)):
        a = torch.randn(m, k, device="cuda", generator=g).to(dtype)
        w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(dtype)
        fn = (lambda: K.gemm(a, w)) if ours else (lambda: torch.matmul(a, w.t()))
        r = rate(fn, 2.0 * m * n * k)
        a.zero_()
        w.zero_()
        z = rate(fn, 2.0 * m * n * k)
        res[name] = {"normal_data": r, "zero_data": z}
        del a, w
    return res

def thread_cpu_seconds() -> dict:
    """user + system CPU seconds of every thread of this process (/proc/self/task/*/stat, fields 14 and 15), keyed "tid:name"."""
    out, tick = {}, os.sysconf("SC_CLK_TCK")
    try:
        for tid in os.listdir("/proc/self/task"):
            with open(f"/proc/self/task/{tid}/stat") as fh:
                head, rest = fh.read().rsplit(")", 1)
            rest = rest.split()
            out[f"{tid}:{head.split('(', 1)[1]}"] = (int(rest[11]) + int(rest[12])) / tick
    except OSError:
        pass
    return out


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={env_world}; they must agree")
    if args.plumbing_only:
        return plumbing_only(args)
    from controlanimate_amd import window_shard as WS
    from tools import ab_switches
    switched_off = ab_switches.apply_from_env()  # (A/B runs only: the product has every form on and reads no environment)
    if switched_off:
        print(f"[bench] A/B run, switched off: {sorted(switched_off)}", file=sys.stderr)
    rank, world, local_rank = WS.init_distributed()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    local_rank %= torch.cuda.device_count()  # (ranks may share a GPU in a gloo rehearsal run, CA_DIST_BACKEND=gloo)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    timer = KernelTimer()
    if not args.no_roofline:
        timer.install()
    out = measure(args, timer, rank, world, device, headline=True)
    if (rank == 0 and world == 1 and not args.no_other_configs and args.config == 2 and args.dtype == "fp16" and not args.no_roofline
            and args.frames is None and args.size is None and args.controlnets is None):
        out["other_configs"] = other_configs(args, timer, device)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(usable_cores())
        if not args.no_cpu_baseline_config2:
            out["cpu_baseline"]["config2_step"] = cpu_baseline_config2_cached(usable_cores())
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


OTHER_CONFIGS = (("config1", 1, "fp16"), ("config3", 3, "fp16"), ("config4", 4, "fp16"), ("config5", 5, "fp16"), ("config2_bf16", 2, "bf16"))


def other_configs(args, timer, device):
    """After the headline and OUTSIDE `value`: one short timed pass (5 loop iterations from a window start, after an untimed first
    window that captures the hipGraph) of BASELINE configs 1, 3, 4, 5 and of config 2 in bf16 -- the product's loop, same calls as the
    headline -- so that every configuration's ms per step, frames/s, MFMA fraction and dominant-kernel roofline are witnessed by
    whoever runs the default command.  A configuration that fails reports its error and the others still run."""
    import argparse as _ap
    import gc
    res = {}
    for name, cfg, dt in OTHER_CONFIGS:
        a = _ap.Namespace(**vars(args))
        a.config, a.dtype, a.steps, a.warmup, a.no_vae, a.shapes = cfg, dt, min(5, int(CONFIGS[cfg]["steps"])), 2, True, False
        t0 = time.perf_counter()
        try:
            o = measure(a, timer, 0, 1, device, headline=False)
            roof = o.get("roofline") or {}
            res[name] = {"workload": o["config"]["workload"], "dtype": dt, "steps": a.steps, "ms_per_step": o["ms_per_step"],
                         "frames_per_sec": o["value"], "frames_per_sec_steady_state": o["frames_per_sec_steady_state"],
                         "step_algorithmic_tflop": o["step_algorithmic_tflop"], "step_mfma_frac": o["step_mfma_frac"],
                         "step_mfma_frac_algorithmic": o["step_mfma_frac_algorithmic"], "hip_graph": o["hip_graph"],
                         "graph_replays_in_timed_region": o["graph_replays_in_timed_region"],
                         "roofline": {k: roof.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launches",
                                                                "avg_launch_us", "share_of_step_time")} if roof else None,
                         "wall_s": round(time.perf_counter() - t0, 1)}
        except Exception as exc:  # (never lose the headline line to a secondary configuration)
            res[name] = {"error": f"{type(exc).__name__}: {exc}"[:400]}
        gc.collect()
        torch.cuda.empty_cache()
    return res


def measure(args, timer, rank, world, device, headline=True):
    """One configuration through the product's loop: models, untimed first window (eager step 0 + hipGraph capture), warm-up, the
    timed K steps, the instrumented roofline pass.  Returns the JSON object (the caller adds the CPU baseline)."""
    from controlanimate_amd import kernels as K
    from controlanimate_amd import window_shard as WS
    from controlanimate_amd.context import dispatch
    from controlanimate_amd.controlresiduals_pipeline import MultiControlNetResidualsPipeline
    from controlanimate_amd.schedulers import get_scheduler
    from controlanimate_amd.configs import NOISE_SCHEDULER_KWARGS
    timer.records, timer.other, timer.bytes, timer.enabled = [], [], {}, False
    wl = dict(CONFIGS[args.config])
    custom = False
    if args.frames is not None:
        wl["frames"], custom = args.frames, True
    if args.size is not None:
        wl["height"] = wl["width"] = args.size
        custom = True
    if args.controlnets is not None:
        wl["controlnets"], custom = args.controlnets, True
    base = CONFIGS[args.config]
    scale = (wl["frames"] / base["frames"]) * (wl["height"] * wl["width"]) / (base["height"] * base["width"])  # (attention terms scale slightly faster)
    steps_per_window = wl["steps"]

    dtype = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    unet, nets = build_models(wl, device, dtype, rank)
    arenas = [unet.arena.buffer] + [n.arena.buffer for n in nets]
    bytes_bcast = WS.broadcast_weights(arenas)
    if world > 1:  # the broadcast is load-bearing (ranks > 0 were built with other weights): every rank must now hold rank 0's bytes
        chk = torch.stack([a[:: max(1, a.numel() // (1 << 16))].to(torch.int64).sum() for a in arenas])  # (device tensors: RCCL and gloo both take them)
        lo, hi = chk.clone(), chk.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        assert torch.equal(lo, hi), "ranks disagree on the broadcast weight arenas"

    f, lh, lw = wl["frames"], wl["height"] // 8, wl["width"] // 8
    g = torch.Generator().manual_seed(1234)
    latents = torch.randn(1, 4, f, lh, lw, generator=g).to(device)
    L = 81 if wl["ip"] else 77   # IP-Adapter: 4 image tokens appended to both halves (controlanimation_pipeline.py:698-710)
    pos = (torch.randn(1, L, 768, generator=g) * 0.5).to(device)
    neg = (torch.randn(1, L, 768, generator=g) * 0.5).to(device)
    guidance = wl["guidance"]
    rep = 2 if guidance > 1.0 else 1
    cn_single = wl["guess_mode"] or rep == 1          # ControlNet input selection (:811-813)
    cn = None
    hints_dev = None
    if nets:
        cn = MultiControlNetResidualsPipeline([f"synthetic-canny-{i}" for i in range(len(nets))], [1.0] * len(nets), use_lcm=False,
                                              controlnets=nets, device=device)
        hints_dev = [h for h in torch.rand(f, 3, wl["height"], wl["width"], generator=g).to(device)]  # resident: inputs are in HBM
    sched = get_scheduler(wl["scheduler"], **NOISE_SCHEDULER_KWARGS)

    # THE PRODUCT'S LOOP: every step below is an iteration of ControlAnimationPipeline.__call__ (the reference's
    # controlanimation_pipeline.py:790-855) -- latents in, latents out, no VAE / text encoder (output_type="latent",
    # prompt embeddings given).  One call = steps [lo, hi) of one window (`step_range`, the pipeline's own hook); a call that
    # starts at step 0 is a window start and does the window's work: prompt embeddings and control frames copied into the
    # tensors the captured hipGraph reads, text / IP K/V and hint embeddings recomputed in place, the sampler noise of the
    # window drawn on the host and uploaded asynchronously.  The hipGraph is the pipeline's default (use_hip_graph): captured
    # once in the first window, replayed for every later step of every later window.
    from controlanimate_amd.controlanimation_pipeline import ControlAnimationPipeline
    pipe = ControlAnimationPipeline(vae=None, text_encoder=None, tokenizer=None, unet=unet, scheduler=sched).to(device)
    pipe.use_hip_graph = not args.no_graph
    pipe.overlap_controlnet = not args.no_overlap
    pipe.fuse_controlnet_adds = not args.no_fuse_adds
    pipe.window_graph = bool(args.window_graph)
    per_step_events = not (args.window_graph and not args.no_graph)  # (a callback needs the host between steps: it selects the per-step path)
    if args.steps_in_flight is not None:
        pipe.steps_in_flight = args.steps_in_flight
    if args.pace_wait is not None:
        pipe.pace_wait = args.pace_wait
    lat0 = latents * float(getattr(sched, "init_noise_sigma", 1.0))
    gen = torch.Generator(device="cpu").manual_seed(4321)
    state = {"latents": lat0}
    step_events = []

    def run_steps(lo, hi, record_events=False):
        def cb(i, t, lat):
            if record_events:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                step_events.append(e)
        out = pipe(video_length=f, input_frames=None, height=wl["height"], width=wl["width"], num_inference_steps=steps_per_window,
                   strength=1.0, guidance_scale=guidance, generator=gen, latents=lat0, prompt_embeds=pos, negative_prompt_embeds=neg,
                   multicontrolnetresiduals_pipeline=cn, control_images=hints_dev, use_lcm=False, guess_mode=wl["guess_mode"],
                   output_type="latent", step_range=(lo, hi), callback=cb if (per_step_events or timer.enabled) else None).videos
        if record_events and not per_step_events:  # whole-window replay: one event at the end of the call's steps
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            step_events.append(("end", e, hi - lo))
        state["latents"] = out
        state["replays"] = state.get("replays", 0) + int(pipe.graph_replays)
        return out

    def run_k_steps(k, record_events=False):
        """k consecutive loop iterations starting AT a window start: ceil(k / steps_per_window) calls."""
        done = 0
        while done < k:
            n = min(k - done, steps_per_window)
            if record_events:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                step_events.append(("start", e))
            run_steps(0, n, record_events)
            done += n

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    # first window, untimed: the eager step 0 (fills every cache, warms the allocator) and the capture at step 1
    run_steps(0, steps_per_window)
    torch.cuda.synchronize()
    use_graph = bool(pipe.use_hip_graph) and pipe.graph_fallback_reason is None and pipe._graph_state is not None and (pipe._graph_state["graph"] is not None or bool(pipe._graph_state["wgraphs"]))
    if pipe.use_hip_graph and not use_graph:
        print(f"[bench] hipGraph capture failed ({pipe.graph_fallback_reason}); running eagerly", file=sys.stderr)
    # The timed region starts AT a window start: the warm-up steps are the last ones of the previous window.  K timed steps
    # therefore contain ceil(K / steps_per_window) window starts with their per-window work (one per 10 steps at the default
    # K = 10: twice the rate of a 20-step window, i.e. never under-counted).
    if args.warmup > 0:
        w = min(args.warmup, steps_per_window)
        run_steps(steps_per_window - w, steps_per_window)
    torch.cuda.synchronize()
    barrier()
    timer.enabled = False  # per-launch HIP events cost ~7% of a step: they are taken in a second pass
    torch.cuda.synchronize()
    replays_before = state.get("replays", 0)
    t0 = time.perf_counter()
    c0 = time.process_time()
    th0 = thread_cpu_seconds()
    run_k_steps(args.steps, record_events=True)
    host_enqueue = time.perf_counter() - t0  # the calls have returned (a paced pipeline has also slept until its window was done)
    host_cpu = time.process_time() - c0      # CPU time of the process (all threads) over the same calls
    th1 = thread_cpu_seconds()
    host_threads = [t for t in sorted(((th1[k] - th0.get(k, 0.0), k) for k in th1), reverse=True) if t[0] > 0.0][:12]
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms, prev = [], None
    for e in step_events:
        if isinstance(e, tuple) and e[0] == "start":
            prev = e[1]
        elif isinstance(e, tuple):  # ("end", event, steps): a call that ran as one replay -- its mean step
            step_ms.append(prev.elapsed_time(e[1]) / max(1, e[2]))
            prev = e[1]
        else:
            step_ms.append(prev.elapsed_time(e))
            prev = e
    step_ms.sort()
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    replays_timed = state.get("replays", 0) - replays_before
    roof_elapsed = elapsed
    if not args.no_roofline:
        # Instrumented pass: the SAME calls, eager, one stream, with a HIP-event pair around every GEMM/conv launch (events
        # on the launch stream).  Kept out of the headline timing because ~3000 event records per step slow the step by
        # ~7%; the per-kernel durations it reports agree with the rocprofv3 --kernel-trace averages in profiles/.
        pipe.use_hip_graph, pipe.overlap_controlnet = False, False
        timer.enabled = True
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_steps(0, min(args.steps, 5, steps_per_window))
        torch.cuda.synchronize()
        roof_elapsed = time.perf_counter() - t1
        timer.enabled = False
        pipe.use_hip_graph, pipe.overlap_controlnet = not args.no_graph, not args.no_overlap
    chains_out = None
    if headline and world == 1 and args.chains > 1 and use_graph and not args.window_graph:
        # Throughput mode, beside `value` and never part of it: `chains` independent windows in flight on this GPU -- each its own pipeline object
        # (sampler, captured graph, static buffers) on its own stream and host thread over the same models.  Whole windows from a window start,
        # 2 windows per chain timed after one priming window each.
        from controlanimate_amd.chains import ChainSet
        def job(seed):
            return dict(video_length=f, input_frames=None, height=wl["height"], width=wl["width"], num_inference_steps=steps_per_window, strength=1.0,
                        guidance_scale=guidance, generator=torch.Generator(device="cpu").manual_seed(seed), latents=lat0, prompt_embeds=pos,
                        negative_prompt_embeds=neg, control_images=hints_dev, use_lcm=False, guess_mode=wl["guess_mode"], output_type="latent")
        try:
            cs = ChainSet(pipe, cn, chains=args.chains)
            cs.map([job(10 + k) for k in range(args.chains)], timeout_s=300)          # priming: every chain captures its graph, one at a time
            torch.cuda.synchronize()
            tc1 = time.perf_counter()
            res = cs.map([job(40 + k) for k in range(2 * args.chains)], prime=False, timeout_s=300)  # two windows per chain, all concurrent
            torch.cuda.synchronize()
            tcc = time.perf_counter() - tc1
            ok = all(torch.isfinite(o.videos).all().item() for o in res)
            steps_done = 2 * args.chains * steps_per_window
            chains_out = {"chains": args.chains, "windows": 2 * args.chains, "ms_per_step_equivalent": round(1e3 * tcc / steps_done, 3),
                          "frames_per_sec": round(2 * args.chains * f / tcc, 4), "finite": bool(ok),
                          "replays_per_window": [int(p_.graph_replays) for p_ in cs.pipes],
                          "note": "independent windows in flight on one GPU, one pipeline object per chain over shared models; a window's latency "
                                  "is `chains` times the single-chain one; not part of `value`"}
            del cs, res
        except Exception as exc:  # (never lose the headline to the side measurement)
            chains_out = {"error": f"{type(exc).__name__}: {exc}"[:300]}
    vae_ms = None
    if not args.no_vae and world == 1:  # single-GPU runs only: the other ranks of a scaling run must not wait for it
        vae_ms = time_vae(wl, device, dtype)
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
    if not torch.isfinite(state["latents"]).all():
        raise SystemExit("non-finite latents after the timed region")

    sec_per_step = elapsed / args.steps
    fps = world * f / (steps_per_window * sec_per_step)
    # per ControlNet: the reference feeds it b*f images, or f in guess mode / without CFG
    cn_tflop = wl["cn_tflop"] * scale
    step_tflop = wl["unet_tflop"] * scale + len(nets) * cn_tflop
    shared_tflop = 0.0
    if rep == 2 and dispatch.cfg_shared:
        rows_half, n_tok = f * (wl["height"] // 8) * (wl["width"] // 8), (wl["height"] // 8) * (wl["width"] // 8)
        one = (2 * 2.0 * rows_half * 320 * 2880 + 2.0 * rows_half * 320 * 1600 + 4.0 * n_tok * n_tok * 320 * f) * 1e-12
        shared_tflop = one * (1 + (0 if cn_single else len(nets)))  # UNet + every ControlNet that sees both halves
    # non-guess CFG with an even frame count: the reference's prompt tiling makes the two halves of the ControlNet's batch the same
    # problem end to end (controlnet.py forward_body) -- each net runs on one half: half of its algorithmic work is not executed
    cn_dedup = bool(nets) and rep == 2 and not cn_single and dispatch.cn_cfg_dedup and f % 2 == 0
    if cn_dedup:
        shared_tflop = shared_tflop - one * len(nets) + 0.5 * cn_tflop * len(nets)
    f_new = f - wl["overlap"]
    out = {
        "metric": "frames_per_sec", "value": round(fps, 4), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * sec_per_step, 3),
        "sec_per_denoise_step": round(sec_per_step, 5), "ms_per_step_median_hipevent": round(median_ms, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": wl["name"] + (" [size/frames/controlnets overridden on the command line]" if custom else "") + "; one window per GPU",
                   "baseline_config": args.config, "frames_per_window": f, "height": wl["height"], "width": wl["width"],
                   "steps_per_window": steps_per_window, "scheduler": wl["scheduler"], "guidance_scale": guidance, "cfg_batch": rep,
                   "controlnets": len(nets), "controlnet_batch": (f if cn_single else rep * f) if nets else 0, "guess_mode": wl["guess_mode"],
                   "ip_adapter_tokens": 4 if wl["ip"] else 0, "overlap_length": wl["overlap"],
                   "parallelism": f"window-shard x{world}" if world > 1 else "single GPU",
                   "weight_broadcast_bytes": bytes_bcast},
        # steady state of the sliding window (scripts/vid2vid.py:168-189): a window re-feeds `overlap_length` frames, so it
        # contributes frame_count - overlap_length NEW frames; equal to `value` when the config has no overlap
        "frames_per_sec_steady_state": round(world * f_new / (steps_per_window * sec_per_step), 4),
        "window_starts_in_timed_region": (args.steps + steps_per_window - 1) // steps_per_window,
        "timed_through": "ControlAnimationPipeline.__call__ (latents in / latents out, step_range windows)",
        "per_window_work_in_timed_region": ("prompt / control-frame rebind, hint embedding + text/IP K/V recomputed in place, sampler noise drawn and "
                                            "uploaded, at every window start; the timed region begins at a window start" if use_graph else
                                            "eager run: nothing is cached outside the timed region except across the steps of a window"),
        "graph_replays_in_timed_region": int(replays_timed) if use_graph else 0,
        "window_graph": bool(use_graph and pipe.window_graph and pipe.window_graph_fallback_reason is None and pipe.window_replays > 0),
        "window_graph_fallback_reason": pipe.window_graph_fallback_reason,
        # host side of the product loop per step.  `host_cpu_ms_per_step` = CPU time of the process (all threads) while the K calls
        # ran: what N ranks on one host compete for.  `host_enqueue_ms_per_step` = wall time until the calls returned, before the
        # device synchronise: a replay of the captured graph waits for the previous replay of the same graph (measured: 2.3 ms on
        # an idle queue, ~12 ms behind a running step), so this is mostly waiting, not work
        "host_cpu_ms_per_step": round(1e3 * host_cpu / args.steps, 3),
        "host_enqueue_ms_per_step": round(1e3 * host_enqueue / args.steps, 3),
        "host_cpu_ms_per_step_by_thread": {k.split(":", 1)[1] + "#" + k.split(":", 1)[0]: round(1e3 * d / args.steps, 3) for d, k in host_threads},  # every thread that used CPU (at most 12; /proc granularity 10 ms over the region)
        # CPU time the process total holds beyond every thread alive at the end of the region (threads that exited inside it)
        "host_cpu_ms_per_step_unlisted": round(1e3 * (host_cpu - sum(th1[k] - th0.get(k, 0.0) for k in th1)) / args.steps, 3),
        "host_intra_op_threads": {"inside_call": 1 if pipe.single_host_thread else torch.get_num_threads(), "process": torch.get_num_threads()},
        "steps_in_flight": int(pipe.steps_in_flight), "pace_wait": str(pipe.pace_wait),
        "step_algorithmic_tflop": round(step_tflop, 2),
        # utilisation of the dense MFMA peak by the work that was EXECUTED (the shared CFG prefix runs once: see below);
        # `step_mfma_frac_algorithmic` divides the reference's count (both halves) by the same time and overstates it
        "step_mfma_frac": round((step_tflop - shared_tflop) / sec_per_step / PEAK_MFMA_TFLOPS, 4),
        "step_mfma_frac_algorithmic": round(step_tflop / sec_per_step / PEAK_MFMA_TFLOPS, 4),
        # classifier-free guidance repeats ONE latent tensor for both batch halves: up to the first cross-attention (conv_in,
        # first resnet, first transformer's GroupNorm / proj_in / 4096-token self-attention) the halves are the same
        # computation, which runs once (bit-identical results: tests/test_workload_configs_gpu.py; CA_CFG_SHARED=0 disables it for an A/B run).  The
        # algorithmic count above is the reference's, which computes both halves; this is what was executed.
        "cfg_shared_prefix": bool(shared_tflop > 0),
        "controlnet_cfg_halves_deduplicated": cn_dedup,
        "step_executed_tflop": round(step_tflop - shared_tflop, 2),
        "vae": None if vae_ms is None else {
            **vae_ms,
            "frames_per_sec_end_to_end": round(world * f / (steps_per_window * sec_per_step + 1e-3 * (vae_ms["encode_ms_per_window"] + vae_ms["decode_ms_per_window"])), 4),
            "note": "SD1.5 AutoencoderKL on the same kernels, random weights; encode + decode of all frames of one window "
                    "(brackets the denoise steps; not part of `value`)"},
        "hip_graph": bool(use_graph),
        "two_chains": chains_out,
        "controlnet_second_stream": bool(not args.no_overlap and nets),
    }
    if not args.no_roofline:
        agg = timer.summary()
        # Winograd route (round 5): the convolutions that take it execute 16 multiply-adds per 2 x 2 outputs instead of 36.  The
        # algorithmic count (the reference's direct convolution) stays what throughput is quoted on; the EXECUTED count -- what the
        # matrix pipes actually did -- is lower by 20 / 36 of those launches' direct FLOPs, and step_mfma_frac follows it.
        wino = agg.get("conv3x3_wino_pq256x320")
        if wino and wino["launches"]:
            steps_instr = min(args.steps, 5, steps_per_window)
            wino_direct_tflop = wino["flops"] / steps_instr * 1e-12
            out["winograd"] = {"launches_per_step": round(wino["launches"] / steps_instr, 1), "direct_tflop_per_step": round(wino_direct_tflop, 3),
                               "executed_tflop_per_step": round(wino_direct_tflop * 16.0 / 36.0, 3)}
            executed = step_tflop - shared_tflop - wino_direct_tflop * 20.0 / 36.0
            out["step_executed_tflop"] = round(executed, 2)
            out["step_mfma_frac"] = round(executed / sec_per_step / PEAK_MFMA_TFLOPS, 4)
        if agg:
            dom = max(agg, key=lambda k: agg[k]["ms"])
            d = agg[dom]
            rname = rocprof_kernel_name(dom, args.dtype)
            traffic, traffic_src = pmc_traffic(rname, "config%d" % args.config if not custom else "custom", args.dtype)
            kname = kernel_display_name(dom)
            # which roof bounds this instantiation's launch mix: algorithmic FLOP per algorithmic byte (every operand
            # once) against the machine balance 2500 TFLOP/s : 8 TB/s = 312 FLOP/B
            gbps = d["bytes"] / (d["ms"] * 1e-3) / 1e9 if d["ms"] > 0 else 0.0
            hbm_bound = d["bytes"] > 0 and d["flops"] / d["bytes"] < PEAK_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBPS * 1e9)
            mfma_view = {"achieved": round(d["tflops"], 2), "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(d["tflops"] / PEAK_MFMA_TFLOPS, 4)}
            hbm_view = {"achieved": round(gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(gbps / PEAK_HBM_GBPS, 4)}
            out["roofline"] = {"bound": "hbm" if hbm_bound else "mfma", "kernel": kname, "rocprof_name": rname + ("" if rname.endswith(">") else "...>") if rname else None,
                               **(hbm_view if hbm_bound else mfma_view),
                               "other_roof": mfma_view if hbm_bound else hbm_view,
                               "traffic": traffic,
                               "traffic_unit": ("HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/%s: same workload and dtype)" % traffic_src)
                               if traffic is not None else "null: no committed PMC summary for this workload / dtype",
                               "launches": d["launches"], "avg_launch_us": round(d["avg_us"], 2),
                               "flop_per_launch": round(d["flops"] / d["launches"], 1),
                               "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"], 1),
                               "flop_per_byte": round(d["flops"] / d["bytes"], 1) if d["bytes"] else None,
                               "share_of_step_time": round(d["ms"] * 1e-3 / roof_elapsed, 4),
                               "measured": "HIP events around every launch, instrumented pass of %d steps after the timed region" % min(args.steps, 5)}
            # per family: algorithmic FLOPs and bytes (every operand once) per launch, so that any family's traffic ratio can
            # be recomputed against the PMC summary in profiles/, not only the dominant one's
            out["kernel_family"] = {k: {"launches": v["launches"], "avg_us": round(v["avg_us"], 2), "tflops": round(v["tflops"], 2),
                                        "algorithmic_bytes_per_launch": round(v["bytes"] / v["launches"], 1),
                                        "algorithmic_gbps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else 0.0,
                                        "rocprof_name": rocprof_kernel_name(k, args.dtype) or None,
                                        "time_share": round(v["ms"] * 1e-3 / roof_elapsed, 4)} for k, v in sorted(agg.items())}
    if headline and not args.no_roofline and world == 1 and "roofline" in out:
        out["roofline"]["matrix_rate_vs_operand_data"] = matrix_rate_vs_operand_data(dtype)
    if args.shapes and rank == 0 and not args.no_roofline:
        for row in timer.by_shape():
            print(row, file=sys.stderr)
        for row in timer.other_summary():
            print(row, file=sys.stderr)
    # let go of everything this configuration holds on the device (the captured graph pins the models and their arenas)
    pipe.release_graph()
    timer.records, timer.other, timer.bytes = [], [], {}
    return out


if __name__ == "__main__":
    main()
