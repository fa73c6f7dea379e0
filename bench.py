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

STEPS_PER_WINDOW = 20
PEAK_MFMA_TFLOPS = 2500.0  # dense bf16/fp16 MFMA, MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--size", type=int, default=512, help="frame height = width in pixels")
    ap.add_argument("--controlnets", type=int, default=1)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--shapes", action="store_true", help="also print the per-shape GEMM/conv table (stderr)")
    ap.add_argument("--no-vae", action="store_true", help="skip the VAE encode/decode timing (reported beside the metric)")
    ap.add_argument("--no-overlap", action="store_true", help="run the ControlNet stack on the main stream (no 2nd-stream overlap)")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch every kernel eagerly.  Default: the ControlNet + UNet part of a step is captured once as a "
                         "hipGraph (both streams) and replayed -- same kernels, bit-identical results "
                         "(tests/test_graph_gpu.py), ~1 ms instead of ~50 ms of host time per step, so the loop stays "
                         "GPU-bound when N ranks share one host; falls back to eager if the capture fails")
    ap.add_argument("--graph", action="store_true", help=argparse.SUPPRESS)  # (old spelling of the default)
    ap.add_argument("--plumbing-only", action="store_true",
                    help="multi-rank plumbing rehearsal WITHOUT the hot path (runs on a CPU box over gloo): rendezvous, "
                         "weight-arena broadcast, barriers, max-over-ranks timing, rank-0 JSON with `dry_run: true` and "
                         "`value: null`.  Never a measurement; used by tests/test_bench_launch.py")
    return ap.parse_args()


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
    out0, _ = procs[0].communicate()
    rc = procs[0].returncode
    for pr in procs[1:]:
        try:
            pr.wait(timeout=120 if rc == 0 else 5)
        except subprocess.TimeoutExpired:
            pr.kill()  # exact PID of a child this parent started
            pr.wait()
        rc = rc or pr.returncode
    sys.stdout.write(out0 or "")
    sys.stdout.flush()
    return rc


def randomize_zero_init_(model, std=0.02, seed=0):
    """Real checkpoints are non-zero where the architecture zero-initialises (motion proj_out,
    ControlNet zero-convs): all-zero operands would also let the chip clock higher (DVFS)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    for p in model.parameters():
        if p.numel() and float(p.detach().abs().max()) == 0.0 and p.dim() > 1:
            p.data.copy_((torch.randn(p.shape, generator=g) * std).to(p.device))


def tile_name(m: int, n: int, k: int, conv: bool) -> str:
    """Mirrors launch_gemm() / splitk_plan() in csrc/ca_gemm.hip (DMA path, default knobs): which k_gemm_dma
    instantiation a launch runs.  The label is the BMxBN block tile (+ `_splitk` for the slab schedule)."""
    def cdiv(a, b):
        return (a + b - 1) // b
    if k % 64:  # register-staged fallback kernel (conv_in, hint embedding)
        wide = n % 128 == 0 and cdiv(m, 128) * cdiv(n, 128) >= 512
        return "reg_128x128" if wide else "reg_128x64"
    if conv and n % 128 == 0 and cdiv(m, 128) * (n // 128) < 384 and k // 64 >= 48:
        return "128x128_splitk"
    if (not conv) and n >= 5120 and k >= 640 and n % 128 == 0 and cdiv(m, 256) * (n // 128) >= 512:
        return "256x128"
    wide = n % 128 == 0 and cdiv(m, 128) * cdiv(n, 128) >= 512
    if not wide and n % 160 == 0 and cdiv(m, 128) * (n // 160) >= 512:
        return "128x160"
    return "128x128" if wide else "128x64"


def rocprof_kernel_name(family: str, dtype: str) -> str:
    """The demangled kernel name rocprofv3 reports for a family label (profiles/*kernel_stats.csv)."""
    op, tile = family.split("_", 1)
    if tile.startswith("reg_"):
        return ""
    dt = 1 if dtype == "fp16" else 0
    bm, bn = tile.replace("_splitk", "").split("x")
    waves = "4, 2" if tile == "256x128" else ("4, 1" if tile == "128x64" else "2, 2")
    return f"k_gemm_dma<{dt}, {bm}, {bn}, {waves}, {1 if op == 'conv3x3' else 0}, "


def pmc_traffic(kernel_prefix: str):
    """HBM-side bytes per launch of a kernel from the committed PMC summary (profiles/round1_pmc_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this same command, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  None when the summary has no entry."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "round1_pmc_traffic.json")
    if not kernel_prefix or not os.path.exists(path):
        return None
    with open(path) as fh:
        table = json.load(fh)["kernels"]
    for name, row in table.items():
        if name.startswith(kernel_prefix):
            return row["hbm_bytes_per_launch"]
    return None


class KernelTimer:
    """HIP-event timing of every ca_gemm / ca_conv3x3 launch (events are recorded on the stream the
    kernels are launched on: torch's current stream) + the algorithmic FLOPs of each launch."""

    def __init__(self):
        self.records = []  # (variant, flops, start_event, end_event, shape)
        self.other = []
        self.enabled = False

    def install(self):
        from controlanimate_amd import kernels as K
        gemm0, conv0 = K.gemm, K.conv3x3
        timer = self

        def gemm(a, w, **kw):
            if not timer.enabled:
                return gemm0(a, w, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = gemm0(a, w, **kw)
            e.record()
            m, n, k = a.shape[0], w.shape[0], w.shape[1]
            timer.records.append((f"gemm_{tile_name(m, n, k, False)}", 2.0 * m * n * k, s, e, (m, n, k)))
            return out

        def conv3x3(x, w, **kw):
            if not timer.enabled:
                return conv0(x, w, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = conv0(x, w, **kw)
            e.record()
            n = w.shape[0]
            mrows = out.shape[0] * out.shape[1] * out.shape[2]
            timer.records.append((f"conv3x3_{tile_name(mrows, n, 9 * w.shape[3], True)}", 2.0 * mrows * n * 9 * w.shape[3], s, e,
                                  (mrows, n, 9 * w.shape[3])))
            return out

        K.gemm, K.conv3x3 = gemm, conv3x3

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

    def by_shape(self, top=25):
        agg = {}
        for name, flops, s, e, shape in self.records:
            d = agg.setdefault((name.split("_")[0],) + shape, dict(n=0, flops=0.0, ms=0.0))
            d["n"] += 1
            d["flops"] += flops
            d["ms"] += s.elapsed_time(e)
        rows = sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:top]
        return [dict(op=k[0], m=k[1], n=k[2], k=k[3], launches=v["n"], ms=round(v["ms"], 3), avg_us=round(1e3 * v["ms"] / v["n"], 1),
                     tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)) for k, v in rows]

    def summary(self):
        agg = {}
        for name, flops, s, e, _shape in self.records:
            d = agg.setdefault(name, dict(launches=0, flops=0.0, ms=0.0))
            d["launches"] += 1
            d["flops"] += flops
            d["ms"] += s.elapsed_time(e)
        for d in agg.values():
            d["avg_us"] = 1e3 * d["ms"] / d["launches"]
            d["tflops"] = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
        return agg


def build_models(args, device, dtype):
    from controlanimate_amd.configs import controlnet_config, unet_config
    from controlanimate_amd.controlnet import ControlNetModel
    from controlanimate_amd.unet import UNet3DConditionModel
    torch.manual_seed(0)
    with torch.device(device):
        unet = UNet3DConditionModel.from_config(unet_config("v2"))
        nets = [ControlNetModel.from_config(controlnet_config()) for _ in range(args.controlnets)]
    randomize_zero_init_(unet, seed=1)
    for i, n in enumerate(nets):
        randomize_zero_init_(n, seed=2 + i)
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


def time_vae(args, device, dtype):
    """ms per window for encoding / decoding all frames with the HIP AutoencoderKL."""
    from controlanimate_amd.vae import AutoencoderKL
    torch.manual_seed(0)
    vae = AutoencoderKL.from_config().to(device).prepare(device, dtype)
    g = torch.Generator().manual_seed(4321)
    imgs = (torch.rand(args.frames, 3, args.size, args.size, generator=g) * 2 - 1).to(device)
    lat = torch.randn(args.frames, 4, args.size // 8, args.size // 8, generator=g).to(device)
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
              for _ in range(1 + args.controlnets)]
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


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={env_world}; they must agree")
    if args.plumbing_only:
        return plumbing_only(args)
    from controlanimate_amd import kernels as K
    from controlanimate_amd import window_shard as WS
    from controlanimate_amd.controlresiduals_pipeline import MultiControlNetResidualsPipeline
    from controlanimate_amd.schedulers import DiffusersLCMScheduler
    from controlanimate_amd.configs import NOISE_SCHEDULER_KWARGS

    rank, world, local_rank = WS.init_distributed()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    local_rank %= torch.cuda.device_count()  # (ranks may share a GPU in a gloo rehearsal run, CA_DIST_BACKEND=gloo)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dtype = torch.float16 if args.dtype == "fp16" else torch.bfloat16

    timer = KernelTimer()
    if not args.no_roofline:
        timer.install()
    unet, nets = build_models(args, device, dtype)
    bytes_bcast = WS.broadcast_weights([unet.arena.buffer] + [n.arena.buffer for n in nets])

    f, hw = args.frames, args.size // 8
    g = torch.Generator().manual_seed(1234)
    latents = torch.randn(1, 4, f, hw, hw, generator=g).to(device)
    pos = (torch.randn(1, 77, 768, generator=g) * 0.5).to(device)
    neg = (torch.randn(1, 77, 768, generator=g) * 0.5).to(device)
    prompt = torch.cat([neg, pos]).contiguous()
    guidance = 1.1
    cn = None
    if nets:
        cn = MultiControlNetResidualsPipeline([f"synthetic-canny-{i}" for i in range(len(nets))], [1.0] * len(nets), use_lcm=False,
                                              controlnets=nets, device=device)
        hints = torch.rand(f, 3, args.size, args.size, generator=g)
        cn.prep_control_images([h for h in hints], do_classifier_free_guidance=True, guess_mode=False)
    sched = DiffusersLCMScheduler(**NOISE_SCHEDULER_KWARGS)
    sched.set_timesteps(STEPS_PER_WINDOW)
    cpad = unet.conv_in.cin_pad
    noise_dev = torch.randn(1, 4, f, hw, hw, generator=g).to(device)

    state = {"latents": latents}
    x_static = torch.empty((2 * f, hw, hw, cpad), device=device, dtype=dtype)
    t_static = torch.zeros(1, device=device, dtype=torch.float32)
    graph_state = {"graph": None, "eps": None}
    overlap = {"on": not args.no_overlap}

    def model_eps(t):
        """ControlNet residuals + UNet3D eps for the contents of x_static at (device) timestep t."""
        down = mid = None
        if cn is not None:
            if overlap["on"]:  # ControlNet beside the UNet encoder on a second stream (as the pipeline does)
                down = cn.residuals_nhwc_async(x_static, t, prompt, False)
            else:
                down, mid = cn.residuals_nhwc(x_static, t, prompt, False)
        return unet.forward_nhwc(x_static, 2, f, t, prompt, down, mid)

    def step(i):
        idx = i % STEPS_PER_WINDOW
        K.latents_to_nhwc(state["latents"], cpad, 2, sched.input_scale(idx), dtype, out=x_static)
        if graph_state["graph"] is not None:
            t_static.fill_(float(sched.timesteps[idx]))
            graph_state["graph"].replay()
            eps = graph_state["eps"]
        else:
            eps = model_eps(sched.timesteps[idx])
        coef, clip = sched.coefficients(idx)
        state["latents"], _ = K.cfg_scheduler_step(eps, 2, guidance, state["latents"], noise_dev, coef, clip)
        if idx == STEPS_PER_WINDOW - 1:
            state["latents"] = latents  # next window

    def capture_graph():
        model_eps(t_static)  # warm: prompt K/V and hint-embedding caches, allocator pools
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            graph_state["eps"] = model_eps(t_static)
        graph_state["graph"] = gr

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    use_graph = not args.no_graph
    if use_graph:
        try:
            capture_graph()
        except Exception as exc:  # keep measuring: eager launches are the same work
            print(f"[bench] hipGraph capture failed ({type(exc).__name__}: {exc}); running eagerly", file=sys.stderr)
            graph_state["graph"], use_graph = None, False
            torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    barrier()
    timer.enabled = False  # per-launch HIP events cost ~7% of a step: they are taken in a second pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    roof_elapsed = elapsed
    if not args.no_roofline:
        # Instrumented pass: the SAME launches, eager, with a HIP-event pair around every GEMM/conv launch
        # (events on the launch stream).  Kept out of the headline timing because ~3000 event records
        # per step slow the step by ~7% (84 -> 90 ms); the per-kernel durations it reports agree with
        # the rocprofv3 --kernel-trace averages in profiles/.
        gr, graph_state["graph"] = graph_state["graph"], None
        overlap["on"] = False  # one stream: a launch's event pair then times that kernel alone
        timer.enabled = True
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(min(args.steps, 5)):
            step(i)
        torch.cuda.synchronize()
        roof_elapsed = time.perf_counter() - t1
        timer.enabled = False
        graph_state["graph"] = gr
    vae_ms = None
    if not args.no_vae and world == 1:  # single-GPU runs only: the other ranks of a scaling run must not wait for it
        vae_ms = time_vae(args, device, dtype)
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
    if not torch.isfinite(state["latents"]).all():
        raise SystemExit("non-finite latents after the timed region")

    sec_per_step = elapsed / args.steps
    fps = world * f / (STEPS_PER_WINDOW * sec_per_step)
    unet_tflop = 35.53 * (f / 16) * (args.size / 512) ** 2  # BASELINE.md section 3 (config 2; attention terms scale slightly faster)
    cn_tflop = 9.07 * (f / 16) * (args.size / 512) ** 2
    step_tflop = unet_tflop + len(nets) * cn_tflop
    out = {
        "metric": "frames_per_sec", "value": round(fps, 4), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * sec_per_step, 3),
        "sec_per_denoise_step": round(sec_per_step, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"SD1.5+mm_sd_v15_v2 UNet3D, {f} frames {args.size}x{args.size}, {STEPS_PER_WINDOW} LCM steps/window, "
                               f"CFG g={guidance} (batch 2), {len(nets)} ControlNet(s); one window per GPU",
                   "frames_per_window": f, "steps_per_window": STEPS_PER_WINDOW, "controlnets": len(nets),
                   "parallelism": f"window-shard x{world}" if world > 1 else "single GPU",
                   "weight_broadcast_bytes": bytes_bcast},
        "step_algorithmic_tflop": round(step_tflop, 2),
        "step_mfma_frac": round(step_tflop / sec_per_step / PEAK_MFMA_TFLOPS, 4),
        "vae": None if vae_ms is None else {
            **vae_ms,
            "frames_per_sec_end_to_end": round(world * f / (STEPS_PER_WINDOW * sec_per_step + 1e-3 * (vae_ms["encode_ms_per_window"] + vae_ms["decode_ms_per_window"])), 4),
            "note": "SD1.5 AutoencoderKL on the same kernels, random weights; encode + decode of all frames of one window "
                    "(brackets the 20 denoise steps; not part of `value`)"},
        "hip_graph": bool(use_graph),
        "controlnet_second_stream": bool(not args.no_overlap and nets),
    }
    if not args.no_roofline:
        agg = timer.summary()
        if agg:
            dom = max(agg, key=lambda k: agg[k]["ms"])
            d = agg[dom]
            rname = rocprof_kernel_name(dom, args.dtype)
            out["roofline"] = {"bound": "mfma", "kernel": f"k_gemm_dma<{dom}>", "rocprof_name": rname + "...>" if rname else None,
                               "achieved": round(d["tflops"], 2), "peak": PEAK_MFMA_TFLOPS,
                               "unit": "TFLOP/s", "frac": round(d["tflops"] / PEAK_MFMA_TFLOPS, 4), "traffic": pmc_traffic(rname),
                               "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/round1_pmc_traffic.json)",
                               "launches": d["launches"], "avg_launch_us": round(d["avg_us"], 2),
                               "flop_per_launch": round(d["flops"] / d["launches"], 1),
                               "share_of_step_time": round(d["ms"] * 1e-3 / roof_elapsed, 4),
                               "measured": "HIP events around every launch, instrumented pass of %d steps after the timed region" % min(args.steps, 5)}
            out["kernel_family"] = {k: {"launches": v["launches"], "avg_us": round(v["avg_us"], 2), "tflops": round(v["tflops"], 2),
                                        "time_share": round(v["ms"] * 1e-3 / roof_elapsed, 4)} for k, v in sorted(agg.items())}
    if args.shapes and rank == 0 and not args.no_roofline:
        for row in timer.by_shape():
            print(row, file=sys.stderr)
        for row in timer.other_summary():
            print(row, file=sys.stderr)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(usable_cores())
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
