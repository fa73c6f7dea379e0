"""hipGraph capture of one denoise step (ControlNet stack on the second stream + UNet3D) replays bit-identically
to the eager launches (bench.py --graph; the fork/join of residuals_nhwc_async become graph edges)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SMALL = (64, 128, 256, 256)


def test_graph_replay_equals_eager():
    from controlanimate_amd.configs import controlnet_config, unet_config
    from controlanimate_amd.controlnet import ControlNetModel
    from controlanimate_amd.controlresiduals_pipeline import MultiControlNetResidualsPipeline
    from controlanimate_amd.unet import UNet3DConditionModel
    from oracle.controlnet import ControlNetConfig, init_controlnet_weights
    from oracle.unet3d import UNet3DConfig, init_unet3d_weights
    unet = UNet3DConditionModel.from_config(unet_config("v2", block_out_channels=SMALL))
    unet.load_state_dict(init_unet3d_weights(UNet3DConfig.v2(block_out_channels=SMALL), seed=41))
    net = ControlNetModel.from_config(controlnet_config(block_out_channels=SMALL))
    net.load_state_dict(init_controlnet_weights(ControlNetConfig(block_out_channels=SMALL), seed=42))
    unet.to(DEV).prepare(DEV)
    net.to(DEV).prepare(DEV)
    f, hw = 8, 16
    g = torch.Generator().manual_seed(3)
    cn = MultiControlNetResidualsPipeline(["c"], [0.9], use_lcm=False, controlnets=[net], device=DEV)
    cn.prep_control_images([h for h in torch.rand(f, 3, 8 * hw, 8 * hw, generator=g)], do_classifier_free_guidance=True, guess_mode=False)
    prompt = (torch.randn(2, 77, 768, generator=g) * 0.5).to(DEV)
    x_static = torch.empty(2 * f, hw, hw, unet.conv_in.cin_pad, device=DEV, dtype=torch.float16)
    t_static = torch.zeros(1, device=DEV)

    def model_eps():
        down = cn.residuals_nhwc_async(x_static, t_static, prompt, False)
        return unet.forward_nhwc(x_static, 2, f, t_static, prompt, down, None)

    xs = [torch.randn(x_static.shape, generator=g).half().to(DEV) for _ in range(2)]
    x_static.copy_(xs[0]); t_static.fill_(500.0)
    model_eps()  # warm caches / allocator
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        eps_g = model_eps()
    for x, t in ((xs[0], 500.0), (xs[1], 120.0), (xs[0], 500.0)):
        x_static.copy_(x); t_static.fill_(t)
        gr.replay()
        torch.cuda.synchronize()
        got = eps_g.clone()
        ref = model_eps()
        torch.cuda.synchronize()
        assert torch.isfinite(got).all() and torch.equal(got, ref)


@pytest.mark.parametrize("scenario", ["lcm_cfg_controlnet", "native_lcm_guess"])
def test_pipeline_hip_graph_option_is_bit_identical(scenario):
    """ControlAnimationPipeline(use_hip_graph=True): eager step 0, captured replay afterwards == all-eager."""
    from controlanimate_amd.configs import NOISE_SCHEDULER_KWARGS
    from controlanimate_amd.controlanimation_pipeline import ControlAnimationPipeline
    from controlanimate_amd.controlresiduals_pipeline import MultiControlNetResidualsPipeline
    from controlanimate_amd.schedulers import get_scheduler
    from tests.test_pipeline_gpu import build
    native = scenario == "native_lcm_guess"
    ucfg, uw, unet, ccfg, cws, nets = build("v2", seed=51, n_controlnets=1, **({"time_cond_proj_dim": 256} if native else {}))
    f, hw = 8, 8
    g = torch.Generator().manual_seed(9)
    pos, neg = torch.randn(1, 77, 768, generator=g) * 0.5, torch.randn(1, 77, 768, generator=g) * 0.5
    hints = torch.rand(f, 3, 8 * hw, 8 * hw, generator=g)
    lat_in = torch.randn(1, 4, f, hw, hw, generator=g) * 0.8
    outs = []
    for use_graph in (False, True):
        sched = None if native else get_scheduler("LCMScheduler", **NOISE_SCHEDULER_KWARGS)
        pipe = ControlAnimationPipeline(vae=None, text_encoder=None, tokenizer=None, unet=unet, scheduler=sched).to(DEV)
        pipe.use_hip_graph = use_graph
        cn = MultiControlNetResidualsPipeline(["n0"], [0.8], use_lcm=native, controlnets=nets, device=DEV)
        torch.manual_seed(3)
        steps = []
        out = pipe(video_length=f, input_frames=None, height=8 * hw, width=8 * hw, num_inference_steps=4, strength=0.5 if native else 1.0,
                   guidance_scale=7.5 if native else 1.3, generator=torch.Generator(device="cpu").manual_seed(3),
                   multicontrolnetresiduals_pipeline=cn, prompt_embeds=pos, negative_prompt_embeds=neg, use_lcm=native,
                   guess_mode=native, input_latents=lat_in, control_images={"n0": [h for h in hints]}, output_type="latent",
                   callback=lambda i, t, l: steps.append(l.clone())).videos
        torch.cuda.synchronize()
        assert pipe.graph_replays == (len(steps) - 1 if use_graph else 0)  # the capture really happened
        outs.append((out.clone(), steps))
    assert len(outs[0][1]) == len(outs[1][1]) >= 3
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.isfinite(outs[0][0]).all()


@pytest.mark.parametrize("scenario", ["lcm_cfg_controlnet", "native_lcm_guess"])
def test_pipeline_graph_persists_across_windows(scenario):
    """The product default (use_hip_graph=True): ONE capture serves every later window of the same signature -- each
    window brings new prompt embeddings and new control frames, which are copied into the tensors the captured kernels
    read, the per-window caches (text K/V, hint embedding) are refreshed in place, and every step of windows 2, 3 replays.
    Three windows with the graph == the same three windows eagerly, bit for bit (latents of every step)."""
    from controlanimate_amd.configs import NOISE_SCHEDULER_KWARGS
    from controlanimate_amd.controlanimation_pipeline import ControlAnimationPipeline
    from controlanimate_amd.controlresiduals_pipeline import MultiControlNetResidualsPipeline
    from controlanimate_amd.schedulers import get_scheduler
    from tests.test_pipeline_gpu import build
    native = scenario == "native_lcm_guess"
    ucfg, uw, unet, ccfg, cws, nets = build("v2", seed=61, n_controlnets=1, **({"time_cond_proj_dim": 256} if native else {}))
    f, hw, nsteps = 8, 8, 4
    g = torch.Generator().manual_seed(19)
    windows = [dict(pos=torch.randn(1, 77, 768, generator=g) * 0.5, neg=torch.randn(1, 77, 768, generator=g) * 0.5,
                    hints=torch.rand(f, 3, 8 * hw, 8 * hw, generator=g), lat=torch.randn(1, 4, f, hw, hw, generator=g) * 0.8) for _ in range(3)]
    runs = []
    for use_graph in (False, True):
        sched = None if native else get_scheduler("LCMScheduler", **NOISE_SCHEDULER_KWARGS)
        pipe = ControlAnimationPipeline(vae=None, text_encoder=None, tokenizer=None, unet=unet, scheduler=sched).to(DEV)
        assert pipe.use_hip_graph is True  # the default
        pipe.use_hip_graph = use_graph
        cn = MultiControlNetResidualsPipeline(["n0"], [0.8], use_lcm=native, controlnets=nets, device=DEV)
        torch.manual_seed(3)
        gen = torch.Generator(device="cpu").manual_seed(3)
        steps, replays = [], []
        for w in windows:
            out = pipe(video_length=f, input_frames=None, height=8 * hw, width=8 * hw, num_inference_steps=nsteps, strength=0.5 if native else 1.0,
                       guidance_scale=7.5 if native else 1.3, generator=gen, multicontrolnetresiduals_pipeline=cn, prompt_embeds=w["pos"],
                       negative_prompt_embeds=w["neg"], use_lcm=native, guess_mode=native, input_latents=w["lat"],
                       control_images={"n0": [h for h in w["hints"]]}, output_type="latent",
                       callback=lambda i, t, l: steps.append(l.clone())).videos
            torch.cuda.synchronize()
            steps.append(out.clone())
            replays.append(pipe.graph_replays)
        if use_graph:
            n = len(steps) // 3 - 1  # steps per window
            assert replays == [n - 1, n, n], replays  # window 1: eager step 0 + capture; windows 2, 3: every step replays
            assert pipe.graph_fallback_reason is None
        runs.append(steps)
    assert len(runs[0]) == len(runs[1]) >= 12
    for k, (a, b) in enumerate(zip(runs[0], runs[1])):
        assert torch.equal(a, b), f"latents differ at record {k}"
    # the windows really differ (the refresh is not a no-op)
    per = len(runs[0]) // 3
    assert not torch.equal(runs[0][per - 1], runs[0][2 * per - 1])


@pytest.mark.parametrize("intruder", ["one_step_call", "second_pipeline", "direct_forward"])
def test_graph_survives_an_eager_forward_between_windows(intruder):
    """The captured step reads buffers the models own (prompt copy, text K/V, hint embeddings).  Any eager forward of the
    same models between two windows with ANOTHER prompt tensor replaces those caches -- a one-step call of the same pipeline
    (no graph for a single step), a second pipeline sharing the UNet and ControlNet, or a direct `forward_nhwc` / ControlNet
    call.  The next window must notice (`_graph_owns_model_caches`), capture again and still equal the all-eager run bit for
    bit; before round 4 it replayed kernels that read the freed / stale K/V."""
    from controlanimate_amd.configs import NOISE_SCHEDULER_KWARGS
    from controlanimate_amd.controlanimation_pipeline import ControlAnimationPipeline
    from controlanimate_amd.controlresiduals_pipeline import MultiControlNetResidualsPipeline
    from controlanimate_amd.schedulers import get_scheduler
    from tests.test_pipeline_gpu import build
    ucfg, uw, unet, ccfg, cws, nets = build("v2", seed=71, n_controlnets=1)
    f, hw, nsteps = 8, 8, 4
    g = torch.Generator().manual_seed(29)
    mk = lambda: dict(pos=torch.randn(1, 77, 768, generator=g) * 0.5, neg=torch.randn(1, 77, 768, generator=g) * 0.5,
                      hints=torch.rand(f, 3, 8 * hw, 8 * hw, generator=g), lat=torch.randn(1, 4, f, hw, hw, generator=g) * 0.8)
    windows, other = [mk() for _ in range(3)], mk()

    def call(pipe, cn, w, steps, gen, **kw):
        return pipe(video_length=f, input_frames=None, height=8 * hw, width=8 * hw, num_inference_steps=nsteps, strength=1.0,
                    guidance_scale=1.3, generator=gen, multicontrolnetresiduals_pipeline=cn, prompt_embeds=w["pos"],
                    negative_prompt_embeds=w["neg"], use_lcm=False, guess_mode=False, input_latents=w["lat"],
                    control_images={"n0": [h for h in w["hints"]]}, output_type="latent",
                    callback=lambda i, t, l: steps.append(l.clone()), **kw).videos

    runs = []
    for use_graph in (False, True):
        pipe = ControlAnimationPipeline(vae=None, text_encoder=None, tokenizer=None, unet=unet,
                                        scheduler=get_scheduler("LCMScheduler", **NOISE_SCHEDULER_KWARGS)).to(DEV)
        pipe.use_hip_graph = use_graph
        cn = MultiControlNetResidualsPipeline(["n0"], [0.8], use_lcm=False, controlnets=nets, device=DEV)
        torch.manual_seed(3)
        gen = torch.Generator(device="cpu").manual_seed(3)
        steps, replays = [], []
        for k, w in enumerate(windows):
            steps.append(call(pipe, cn, w, steps, gen).clone())
            replays.append(pipe.graph_replays)
            torch.cuda.synchronize()
            if k == 2:
                break
            # ---- the intruder: an eager forward of the same models with another prompt / other control frames
            junk = []
            if intruder == "one_step_call":
                call(pipe, cn, other, junk, torch.Generator(device="cpu").manual_seed(5), step_range=(1, 2))
            elif intruder == "second_pipeline":
                p2 = ControlAnimationPipeline(vae=None, text_encoder=None, tokenizer=None, unet=unet,
                                              scheduler=get_scheduler("LCMScheduler", **NOISE_SCHEDULER_KWARGS)).to(DEV)
                p2.use_hip_graph = False
                cn2 = MultiControlNetResidualsPipeline(["n0"], [0.8], use_lcm=False, controlnets=nets, device=DEV)
                call(p2, cn2, other, junk, torch.Generator(device="cpu").manual_seed(5))
            else:
                x = torch.randn(2 * f, hw, hw, unet.conv_in.cin_pad, generator=g).half().to(DEV)
                pr = torch.cat([other["neg"], other["pos"]]).to(DEV)
                unet.forward_nhwc(x, 2, f, 500.0, pr, None, None)
                nets[0].forward_body(x, 500.0, pr, torch.cat([other["hints"]] * 2).to(DEV))
            torch.cuda.synchronize()
            # garbage over whatever the allocator handed back: a replay that still reads freed K/V would now see NaN
            trash = [torch.full((1 << 20,), float("nan"), device=DEV, dtype=torch.float16) for _ in range(64)]
            del trash
        if use_graph:
            assert pipe.graph_fallback_reason is None
            assert all(r >= nsteps - 1 for r in replays), replays  # every window still replays (after a re-capture)
        runs.append(steps)
    assert len(runs[0]) == len(runs[1]) >= 12
    for k, (a, b) in enumerate(zip(runs[0], runs[1])):
        assert torch.isfinite(b).all() and torch.equal(a, b), f"latents differ at record {k}"


def test_fused_controlnet_adds_equal_the_separate_adds():
    """The reference's 13 `sample + residual` adds (unet.py:567-576, 584-585) inside the zero convolutions' epilogues
    (residuals_nhwc_async(fuse_images=...)) == 13 separate ca_add_bcast passes, bit for bit with one ControlNet (the zero
    convolution rounds its output to the activation type before the residual operand is added, exactly as the stored
    residual was rounded) through the whole loop."""
    from controlanimate_amd.configs import NOISE_SCHEDULER_KWARGS
    from controlanimate_amd.controlanimation_pipeline import ControlAnimationPipeline
    from controlanimate_amd.controlresiduals_pipeline import MultiControlNetResidualsPipeline
    from controlanimate_amd.schedulers import get_scheduler
    from tests.test_pipeline_gpu import build
    ucfg, uw, unet, ccfg, cws, nets = build("v2", seed=71, n_controlnets=1)
    f, hw = 8, 8
    g = torch.Generator().manual_seed(29)
    pos, neg = torch.randn(1, 77, 768, generator=g) * 0.5, torch.randn(1, 77, 768, generator=g) * 0.5
    hints = torch.rand(f, 3, 8 * hw, 8 * hw, generator=g)
    lat = torch.randn(1, 4, f, hw, hw, generator=g)
    outs = []
    for fuse in (False, True):
        pipe = ControlAnimationPipeline(vae=None, text_encoder=None, tokenizer=None, unet=unet,
                                        scheduler=get_scheduler("DDIMScheduler", **NOISE_SCHEDULER_KWARGS)).to(DEV)
        assert pipe.fuse_controlnet_adds is True  # the default
        pipe.fuse_controlnet_adds = fuse
        cn = MultiControlNetResidualsPipeline(["n0"], [0.8], use_lcm=False, controlnets=nets, device=DEV)
        out = pipe(video_length=f, input_frames=None, height=8 * hw, width=8 * hw, num_inference_steps=3, strength=1.0,
                   guidance_scale=7.5, generator=torch.Generator(device="cpu").manual_seed(3), latents=lat.clone(),
                   multicontrolnetresiduals_pipeline=cn, prompt_embeds=pos, negative_prompt_embeds=neg, use_lcm=False,
                   guess_mode=False, control_images={"n0": [h for h in hints]}, output_type="latent").videos
        torch.cuda.synchronize()
        outs.append(out.clone())
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1])


def test_fused_controlnet_adds_two_nets_within_rounding():
    """Two ControlNets: fused = round(round(s1 z1) + round(round(s0 z0) + skip)), separate = round(skip + round(round(s1 z1)
    + round(s0 z0))) -- the same three terms associated differently, so every element agrees within a few fp16 ulps of
    the largest term (and the two are NOT expected to be bit-identical)."""
    from controlanimate_amd import kernels as K
    from controlanimate_amd.controlresiduals_pipeline import MultiControlNetResidualsPipeline
    from tests.test_pipeline_gpu import build
    ucfg, uw, unet, ccfg, cws, nets = build("v2", seed=72, n_controlnets=2)
    f, hw = 8, 8
    g = torch.Generator().manual_seed(31)
    cn = MultiControlNetResidualsPipeline(["n0", "n1"], [0.8, 0.6], use_lcm=False, controlnets=nets, device=DEV)
    cn.prep_control_images([h for h in torch.rand(f, 3, 8 * hw, 8 * hw, generator=g)], do_classifier_free_guidance=True, guess_mode=False)
    prompt = (torch.randn(2, 77, 768, generator=g) * 0.5).to(DEV)
    x = torch.randn(2 * f, hw, hw, nets[0].conv_in.cin_pad, generator=g).half().to(DEV)
    t = torch.full((1,), 480.0, device=DEV)
    bodies = cn.controlnet.forward_bodies(x, t, prompt, cn.prep_images, cn.cond_scale, False)
    base_d = [(torch.randn(o.shape, generator=g) * 0.7).half().to(DEV) for o in bodies[0][0]]
    base_m = (torch.randn(bodies[0][1].shape, generator=g) * 0.7).half().to(DEV)
    fd, fm = cn.controlnet.finish(bodies, (base_d, base_m))
    sd, sm = cn.controlnet.finish(bodies, None)
    torch.cuda.synchronize()
    assert len(fd) == len(sd) == 12
    d0, m0 = nets[0].apply_zero_convs(*bodies[0][:3], None, bodies[0][3])  # the first net's term alone: the intermediate sums' magnitude
    for a, s_, r, r0 in zip((*fd, fm), (*base_d, base_m), (*sd, sm), (*d0, m0)):
        sep = K.add_bcast(s_, r)
        assert torch.isfinite(a).all() and r.float().abs().max() > 1e-3  # (the random zero convolutions are not zero)
        mags = [s_.float().abs(), r.float().abs(), a.float().abs(), r0.float().abs(), (r.float() - r0.float()).abs(), (r0.float() + s_.float()).abs()]
        scale = torch.stack(mags).amax(0).clamp_min(1e-3)
        ulps = ((a.float() - sep.float()).abs() / (scale * 2.0 ** -10)).max().item()
        assert ulps <= 3.0, ulps


@pytest.mark.parametrize("scenario", ["lcm_cfg_controlnet", "native_lcm_guess", "euler_ancestral_cfg_two_controlnets"])
def test_whole_window_graph_is_bit_identical_to_the_per_step_paths(scenario):
    """ControlAnimationPipeline.window_graph (opt-in; applies when nothing needs the host between steps: no callback): every loop iteration
    of a window -- CFG duplicate with the sampler's input scale, ControlNet stack on the side stream, UNet3D, CFG combine + sampler
    update with the step's coefficients and its slice of the pre-uploaded noise -- is ONE captured hipGraph, replayed once per window
    (/root/reference/animatediff/pipelines/controlanimation_pipeline.py:792-855 has no host dependence between steps either).
    Three windows (new prompts, control frames and latents each; an eager forward with a foreign prompt between windows 2 and 3 forces
    a re-capture) == the same windows on the per-step graph == all-eager, bit for bit; one replay per window."""
    from controlanimate_amd.configs import NOISE_SCHEDULER_KWARGS
    from controlanimate_amd.controlanimation_pipeline import ControlAnimationPipeline
    from controlanimate_amd.controlresiduals_pipeline import MultiControlNetResidualsPipeline
    from controlanimate_amd.schedulers import get_scheduler
    from tests.test_pipeline_gpu import build
    native = scenario == "native_lcm_guess"
    n_nets = 2 if scenario.endswith("two_controlnets") else 1
    sched_name = "EulerAncestralDiscreteScheduler" if n_nets == 2 else "LCMScheduler"
    ucfg, uw, unet, ccfg, cws, nets = build("v2", seed=81, n_controlnets=n_nets, **({"time_cond_proj_dim": 256} if native else {}))
    f, hw, nsteps = 8, 8, 4
    g = torch.Generator().manual_seed(39)
    mk = lambda: dict(pos=torch.randn(1, 77, 768, generator=g) * 0.5, neg=torch.randn(1, 77, 768, generator=g) * 0.5,
                      hints=torch.rand(f, 3, 8 * hw, 8 * hw, generator=g), lat=torch.randn(1, 4, f, hw, hw, generator=g) * 0.8)
    windows, other = [mk() for _ in range(3)], mk()
    names = [f"n{i}" for i in range(n_nets)]
    runs = {}
    for mode in ("eager", "per_step", "window"):
        sched = None if native else get_scheduler(sched_name, **NOISE_SCHEDULER_KWARGS)
        pipe = ControlAnimationPipeline(vae=None, text_encoder=None, tokenizer=None, unet=unet, scheduler=sched).to(DEV)
        assert pipe.use_hip_graph is True and pipe.window_graph is False  # the defaults (window_graph: opt-in, see its comment)
        pipe.use_hip_graph = mode != "eager"
        pipe.window_graph = mode == "window"
        cn = MultiControlNetResidualsPipeline(names, [0.8, 0.5][:n_nets], use_lcm=native, controlnets=nets, device=DEV)
        torch.manual_seed(3)
        gen = torch.Generator(device="cpu").manual_seed(3)
        outs, replays = [], []
        for k, w in enumerate(windows):
            out = pipe(video_length=f, input_frames=None, height=8 * hw, width=8 * hw, num_inference_steps=nsteps, strength=0.5 if native else 1.0,
                       guidance_scale=7.5 if native else 1.3, generator=gen, multicontrolnetresiduals_pipeline=cn, prompt_embeds=w["pos"],
                       negative_prompt_embeds=w["neg"], use_lcm=native, guess_mode=native, input_latents=w["lat"],
                       control_images={n: [h for h in w["hints"]] for n in names}, output_type="latent").videos
            torch.cuda.synchronize()
            outs.append(out.clone())
            replays.append(pipe.graph_replays)
            if k == 1:  # an intruder: the same models, another prompt -- the captured kernels' caches change hands
                pr = torch.cat([other["neg"], other["pos"]]).to(DEV)
                unet.forward_nhwc(torch.randn(2 * f, hw, hw, unet.conv_in.cin_pad, generator=g).half().to(DEV), 2, f, 500.0, pr, None, None)
                torch.cuda.synchronize()
        if mode == "window":
            assert pipe.window_graph_fallback_reason is None and pipe.graph_fallback_reason is None
            assert replays == [1, 1, 1] and pipe.window_replays == 3, (replays, pipe.window_replays)
        elif mode == "per_step":
            assert pipe.window_replays == 0 and all(r >= nsteps - 1 for r in replays), replays
        runs[mode] = outs
    for k in range(3):
        assert torch.isfinite(runs["window"][k]).all()
        assert torch.equal(runs["eager"][k], runs["per_step"][k]), f"per-step graph differs from eager in window {k}"
        assert torch.equal(runs["eager"][k], runs["window"][k]), f"whole-window graph differs from eager in window {k}"
    assert not torch.equal(runs["window"][0], runs["window"][1])


def test_whole_window_graph_steps_aside_when_the_host_is_needed_between_steps():
    """A `callback` (or `record_eps`, a history-carrying sampler, a partial `step_range`) needs the host between steps: such calls keep the
    per-step graph, and a pipeline that alternates the two kinds of call keeps both captures alive."""
    from controlanimate_amd.configs import NOISE_SCHEDULER_KWARGS
    from controlanimate_amd.controlanimation_pipeline import ControlAnimationPipeline
    from controlanimate_amd.schedulers import get_scheduler
    from tests.test_pipeline_gpu import build
    ucfg, uw, unet, ccfg, cws, nets = build("v2", seed=83)
    f, hw = 8, 8
    g = torch.Generator().manual_seed(41)
    pos, neg = torch.randn(1, 77, 768, generator=g) * 0.5, torch.randn(1, 77, 768, generator=g) * 0.5
    lat = torch.randn(1, 4, f, hw, hw, generator=g)
    pipe = ControlAnimationPipeline(vae=None, text_encoder=None, tokenizer=None, unet=unet,
                                    scheduler=get_scheduler("DDIMScheduler", **NOISE_SCHEDULER_KWARGS)).to(DEV)
    pipe.window_graph = True
    kw = dict(video_length=f, input_frames=None, height=8 * hw, width=8 * hw, num_inference_steps=4, strength=1.0, guidance_scale=2.0,
              prompt_embeds=pos, negative_prompt_embeds=neg, use_lcm=False, output_type="latent")
    a = pipe(latents=lat.clone(), **kw).videos.clone()
    assert pipe.window_replays == 1 and pipe.graph_replays == 1
    seen = []
    b = pipe(latents=lat.clone(), callback=lambda i, t, l: seen.append(i), **kw).videos.clone()
    assert pipe.window_replays == 1 and len(seen) == 4 and pipe.graph_replays == 3   # eager step 0 + capture + 3 replays
    c = pipe(latents=lat.clone(), **kw).videos.clone()
    assert pipe.window_replays == 2 and pipe.graph_replays == 1                      # the window capture was kept
    d = pipe(latents=lat.clone(), step_range=(1, 4), **kw).videos                    # (a partial window: per-step)
    assert pipe.window_replays == 2
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(a, c) and torch.isfinite(d).all()
