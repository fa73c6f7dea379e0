"""CPU: host-side logic of the product (no kernels): sampler coefficient tables vs the oracle's
step functions, weight packing layouts, module/key parity with the reference fixtures, IP-Adapter
key renumbering, window planning."""
import json
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def apply_coef(coef, clip, x, eps, noise):
    x0 = (x - coef[0] * eps) * coef[1]
    if clip > 0:
        x0 = x0.clamp(-clip, clip)
    den = coef[2] * x0 + coef[3] * x
    prev = coef[4] * den + coef[5] * eps + (coef[6] * noise if noise is not None else 0)
    return prev, den


@pytest.mark.parametrize("name", ["DDIMScheduler", "LCMScheduler", "EulerDiscreteScheduler", "custom_lcm"])
def test_scheduler_coefficients_equal_oracle_steps(name):
    from controlanimate_amd import schedulers as P
    from oracle import schedulers as O
    kw = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="linear")
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 4, 4, 8, 8, generator=g)
    n_steps = 6
    if name == "custom_lcm":
        p, o = P.LCMScheduler(), O.CustomLCM()
        p.set_timesteps(0.5, 4, 50)
        o.set_timesteps(0.5, 4, 50)
    else:
        p = P.get_scheduler(name, **kw)
        o = {"DDIMScheduler": O.DDIM, "LCMScheduler": O.DiffusersLCM, "EulerDiscreteScheduler": O.EulerDiscrete}[name](**kw)
        p.set_timesteps(n_steps)
        o.set_timesteps(n_steps)
    assert torch.equal(torch.as_tensor(p.timesteps).double(), torch.as_tensor(o.timesteps).double())
    assert abs(float(p.init_noise_sigma) - float(o.init_noise_sigma)) < 1e-5
    xp = xo = x * float(o.init_noise_sigma)
    for i, t in enumerate(o.timesteps):
        eps = torch.randn(x.shape, generator=g)
        noise = torch.randn(x.shape, generator=g)
        sc = p.input_scale(i)
        so = o.scale_model_input(torch.ones(1), t)
        assert abs(sc - float(so)) < 1e-6
        coef, clip = p.coefficients(i)
        xp, _ = apply_coef(coef, clip, xp, eps, noise if p.needs_noise else None)
        if name == "custom_lcm":
            xo, _ = o.step(eps, i, t, xo, noise=noise)
        elif name == "LCMScheduler":
            xo, _ = o.step(eps, t, xo, noise=noise)
        else:
            xo, _ = o.step(eps, t, xo)
        assert torch.allclose(xp, xo, atol=2e-5, rtol=2e-5), (name, i, (xp - xo).abs().max())


def test_lcm_timesteps_known_answers():
    from controlanimate_amd.schedulers import DiffusersLCMScheduler, LCMScheduler
    s = LCMScheduler()
    s.set_timesteps(1.0, 20, 50)
    assert s.timesteps.tolist() == list(range(999, 238, -40))
    s.set_timesteps(0.5, 4, 50)
    assert s.timesteps.tolist() == [499, 379, 259, 139]
    fx = np.load(os.path.join(G, "lcm_custom.npz"))
    for k, (strength, steps) in enumerate(fx["combos"]):
        s.set_timesteps(float(strength), int(steps), 50)
        assert np.array_equal(s.timesteps.numpy(), fx[f"timesteps_{k}"])
    assert np.allclose(s.alphas_cumprod.numpy(), fx["alphas_cumprod"], rtol=0, atol=0)
    d = DiffusersLCMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear")
    d.set_timesteps(20)
    assert d.timesteps.tolist() == list(range(999, 238, -40))


def test_state_dict_keys_and_processor_order_match_reference():
    from controlanimate_amd.configs import unet_config
    from controlanimate_amd.unet import UNet3DConditionModel
    ref = json.load(open(os.path.join(G, "unet3d_keys.json")))
    for ver in ("v1", "v2"):
        with torch.device("meta"):
            m = UNet3DConditionModel.from_config(unet_config(ver))
        mine = {k: list(v.shape) for k, v in m.state_dict().items()}
        # the reference additionally carries 80 never-used block-level to_q/k/v/out tensors
        theirs = {k: v for k, v in ref[ver]["keys"].items() if ".transformer_blocks.0.to_" not in k}
        assert mine == theirs
        assert list(m.attn_processors.keys()) == ref[ver]["attn_processors"]
        assert len(m.attn_processors) == (88 if ver == "v1" else 90)


def test_controlnet_keys_match_oracle_table():
    from controlanimate_amd.configs import controlnet_config
    from controlanimate_amd.controlnet import ControlNetModel
    from oracle.controlnet import ControlNetConfig, controlnet_param_shapes
    with torch.device("meta"):
        m = ControlNetModel.from_config(controlnet_config())
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert mine == controlnet_param_shapes(ControlNetConfig())
    assert sum(int(np.prod(s)) for s in mine.values()) == 361_279_120


def test_weight_packing_layouts_on_cpu():
    from controlanimate_amd.attention import GEGLU
    from controlanimate_amd.attention_processor import Attention
    from controlanimate_amd.layers import HipConv3x3, WeightArena, geglu_interleave
    arena = WeightArena()
    conv = HipConv3x3(4, 16)
    conv.pack(arena, torch.float16)
    geglu = GEGLU(8, 16)
    geglu.pack(arena, torch.float16)
    attn = Attention(32, heads=4, dim_head=8)
    attn.pack(arena, torch.float16)
    xattn = Attention(32, cross_attention_dim=24, heads=4, dim_head=8)
    xattn.pack(arena, torch.float16)
    buf = arena.finalize("cpu")
    assert buf.dtype == torch.uint8 and all(p.offset % 256 == 0 for p in arena.items)
    w = conv.w.t.float()
    assert w.shape == (16, 3, 3, 8)
    assert torch.equal(w[..., :4], conv.weight.detach().permute(0, 2, 3, 1).half().float()) and w[..., 4:].abs().max() == 0
    gw = geglu.w.t.float()
    assert torch.equal(gw[0::2], geglu.proj.weight.detach()[:16].half().float())
    assert torch.equal(gw[1::2], geglu.proj.weight.detach()[16:].half().float())
    assert torch.equal(geglu_interleave(torch.arange(6.0)), torch.tensor([0., 3., 1., 4., 2., 5.]))
    qkv = attn.qkv.t.float()
    assert torch.equal(qkv[32:64], attn.to_k.weight.detach().half().float())
    assert xattn.kv.t.shape == (64, 24) and xattn.to_q.w.t.shape == (32, 32)


def test_ip_adapter_key_renumbering():
    from controlanimate_amd.ip_adapter import IPAdapter
    ref = json.load(open(os.path.join(G, "unet3d_keys.json")))
    keys = ref["v2"]["attn_processors"]
    ip_state = {}
    for n in range(1, 32, 2):                      # SD1.5 ip-adapter checkpoints: layers 1,3,...,31
        ip_state[f"{n}.to_k_ip.weight"] = torch.tensor([float(n)])
        ip_state[f"{n}.to_v_ip.weight"] = torch.tensor([float(n) + 0.5])
    new = IPAdapter.renumber_ip_keys(ip_state, keys)
    slots = [2, 5, 12, 15, 22, 25, 42, 45, 48, 57, 60, 63, 72, 75, 78, 87]   # SURVEY App. C-7 (verified on the reference)
    assert [i for i, k in enumerate(keys) if "attn2" in k] == slots
    assert sorted({int(k.split(".")[0]) for k in new}) == slots
    for j, s in enumerate(slots):
        assert float(new[f"{s}.to_k_ip.weight"]) == 2 * j + 1
    assert keys[slots[-1]].startswith("mid_block")   # order: down, up, then mid


def test_window_plan_and_blend():
    from controlanimate_amd.window_shard import blend_overlap, window_plan, windows_for_rank
    plan = window_plan(40, 16, 8)
    assert plan == [(0, 16), (8, 24), (16, 32), (24, 40)]
    assert window_plan(10, 16, 8) == [(0, 10)]
    assert window_plan(20, 16, 4) == [(0, 16), (12, 20)]
    assert windows_for_rank(7, 1, 4) == [1, 5] and sum(len(windows_for_rank(7, r, 4)) for r in range(4)) == 7
    prev, cur = torch.ones(4, 2, 2), torch.zeros(4, 2, 2)
    out = blend_overlap(prev, cur)
    assert torch.allclose(out[:, 0, 0], torch.tensor([3.5, 2.5, 1.5, 0.5]) / 4)  # Image.blend alpha (n-i-0.5)/n
    with pytest.raises(ValueError):
        window_plan(10, 8, 8)


# ------------------------------------------------------------------------------------------------------------------
# The four samplers added in round 2 (EulerAncestral, LMS, DPM-Solver++ multistep, PNDM): the product's host-computed
# coefficient tables (one fused / one lincomb launch per step) against the step-function-shaped oracle restatements.
def _torch_lincomb(terms):
    out = torch.zeros_like(terms[0][0])
    for x, c in terms:
        out = out + float(c) * x
    return out


@pytest.mark.parametrize("name,oracle_cls,steps", [("EulerAncestralDiscreteScheduler", "EulerAncestral", 7), ("LMSDiscreteScheduler", "LMSDiscrete", 9),
                                                   ("DPMSolverMultistepScheduler", "DPMSolverMultistep", 8), ("DPMSolverMultistepScheduler", "DPMSolverMultistep", 20),
                                                   ("PNDMScheduler", "PNDM", 6), ("PNDMScheduler", "PNDM", 20)])
def test_round2_schedulers_match_their_oracle(name, oracle_cls, steps):
    from controlanimate_amd.configs import NOISE_SCHEDULER_KWARGS
    from controlanimate_amd.schedulers import get_scheduler
    from oracle import schedulers as OS
    prod = get_scheduler(name, **NOISE_SCHEDULER_KWARGS)
    prod.set_timesteps(steps)
    ora = getattr(OS, oracle_cls)(**NOISE_SCHEDULER_KWARGS)
    ora.set_timesteps(steps)
    assert [round(float(t), 3) for t in prod.timesteps] == [round(float(t), 3) for t in ora.timesteps]
    if name == "PNDMScheduler":
        assert len(prod.timesteps) == 12 + steps - 3      # 3 Runge-Kutta steps of 4 evaluations, then the multistep part
    g = torch.Generator().manual_seed(steps)
    shape = (1, 4, 3, 5, 5)
    xp = xo = torch.randn(shape, generator=g, dtype=torch.float64) * float(getattr(prod, "init_noise_sigma", 1.0))
    for i, t in enumerate(prod.timesteps):
        eps = torch.randn(shape, generator=g, dtype=torch.float64)
        noise = torch.randn(shape, generator=g, dtype=torch.float64)
        assert abs(prod.input_scale(i) - float((ora.scale_model_input(torch.ones(1, dtype=torch.float64), ora.timesteps[i]))[0])) < 1e-6
        if getattr(prod, "multistep", False):
            xp = prod.step_device(i, eps, xp, noise, _torch_lincomb)
            xo = ora.step(eps, ora.timesteps[i], xo)[0]
        else:
            c, clip = prod.coefficients(i)
            x0 = (xp - c[0] * eps) * c[1]
            den = c[2] * x0 + c[3] * xp
            xp = c[4] * den + c[5] * eps + c[6] * noise
            xo = ora.step(eps, ora.timesteps[i], xo, noise=noise)[0]
        rel = float((xp - xo).norm() / xo.norm())
        assert rel < 2e-5, (name, i, rel)
    assert torch.isfinite(xp).all()


def test_every_scheduler_of_the_reference_table_is_selectable():
    """modules/controlanimate_pipeline.py:52-61."""
    from controlanimate_amd.configs import NOISE_SCHEDULER_KWARGS
    from controlanimate_amd.schedulers import get_scheduler
    for n in ("EulerDiscreteScheduler", "DDIMScheduler", "DPMSolverMultistepScheduler", "EulerAncestralDiscreteScheduler",
              "LMSDiscreteScheduler", "PNDMScheduler", "LCMScheduler"):
        s = get_scheduler(n, **NOISE_SCHEDULER_KWARGS)
        s.set_timesteps(10)
        assert len(s.timesteps) >= 10
    with pytest.raises(NotImplementedError):
        get_scheduler("UniPCMultistepScheduler")


def test_predrawn_sampler_noise_equals_the_references_per_step_draws():
    """ControlAnimationPipeline draws the sampler noise of a whole window in ONE `torch.randn` (off the per-step critical
    path); the reference draws one tensor per step inside the loop (controlanimation_pipeline.py:1601, global CPU RNG;
    diffusers randn_tensor with a CPU generator).  Same stream, same values -- for element counts that are multiples of 16
    (every real latent shape: 4 x f x h x w); other shapes are drawn step by step by the pipeline."""
    import torch
    for shape in [(1, 4, 16, 64, 64), (1, 4, 8, 32, 32), (1, 4, 8, 8, 8), (1, 4, 16, 64, 96)]:
        assert torch.Size(shape).numel() % 16 == 0
        torch.manual_seed(11)
        seq = [torch.randn(shape) for _ in range(5)]
        torch.manual_seed(11)
        big = torch.randn((5,) + shape)
        assert all(torch.equal(big[i], seq[i]) for i in range(5))
        g = torch.Generator().manual_seed(12)
        seq = [torch.randn(shape, generator=g, dtype=torch.float32) for _ in range(3)]
        g = torch.Generator().manual_seed(12)
        big = torch.randn((3,) + shape, generator=g, dtype=torch.float32)
        assert all(torch.equal(big[i], seq[i]) for i in range(3))


def test_chain_clones_share_models_and_own_their_loop_state():
    """controlanimate_amd/chains.py (two windows in flight per GPU): a chain's pipeline is a new loop over the SAME model objects -- its own sampler
    instance and graph / noise state, the flags of the original; the process-wide intra-op thread switch is taken once around all chains."""
    import torch
    from controlanimate_amd.chains import clone_pipeline, clone_residuals_pipeline, one_host_thread
    from controlanimate_amd.configs import NOISE_SCHEDULER_KWARGS
    from controlanimate_amd.controlanimation_pipeline import ControlAnimationPipeline
    from controlanimate_amd.schedulers import get_scheduler
    unet = object()
    pipe = ControlAnimationPipeline(vae=None, text_encoder=None, tokenizer=None, unet=None, scheduler=get_scheduler("LCMScheduler", **NOISE_SCHEDULER_KWARGS))
    pipe.unet = unet
    pipe.steps_in_flight, pipe.fuse_controlnet_adds, pipe.window_graph = 3, False, True
    twin = clone_pipeline(pipe)
    assert twin is not pipe and twin.unet is unet and twin.scheduler is not pipe.scheduler and type(twin.scheduler) is type(pipe.scheduler)
    assert (twin.steps_in_flight, twin.fuse_controlnet_adds, twin.window_graph, twin.use_hip_graph) == (3, False, True, True)
    assert twin._graph_state is None and twin._noise_state is None
    assert clone_residuals_pipeline(None) is None
    n = torch.get_num_threads()
    with one_host_thread([pipe, twin]):
        assert torch.get_num_threads() == 1 and not pipe.single_host_thread and not twin.single_host_thread
    assert torch.get_num_threads() == n and pipe.single_host_thread and twin.single_host_thread
