"""GPU: parity and properties at the REAL model width.

* BASELINE config 1 shape (SD1.5 + mm v1, 8 frames 256x256, CFG batch 2, 77 tokens) at full width
  against the fp32 oracle on the host -- the same probe BASELINE.md section 2 timed on the reference.
* BASELINE config 2 size (mm v2, 16 frames 512x512, CFG batch 2, 1 ControlNet):
  - ONE full denoise step -- ControlNet residuals + UNet3D eps at (2,4,16,64,64), full width, seeded weights -- against the
    fp32 oracle on the host cores (~100 s, ~40 GB of host memory).  This is the only network-level comparison in which
    the kernels the headline shapes dispatch to actually run (the weight-resident K = 320 kernel, the 128x320 tiles, the
    single-buffer 128x128 / 128x160 tiles at >= 512 blocks, split-K, the 4096-token attention, one-launch GroupNorm,
    producer -> consumer LayerNorm row sums; tests/test_dispatch_plan.py pins which shape takes which);
  - size-independent properties: bit-determinism, CFG-half consistency (identical halves in -> identical halves out),
    ControlNet residual broadcast equivalence (b=1 residuals == the same residuals tiled to b=2), finiteness.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item()


def test_config1_full_width_unet_eps_vs_oracle():
    from controlanimate_amd.configs import unet_config
    from controlanimate_amd.unet import UNet3DConditionModel
    from oracle.unet3d import UNet3DConfig, init_unet3d_weights, unet3d_forward
    cfg = UNet3DConfig.v1()
    w = init_unet3d_weights(cfg, seed=0)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 4, 8, 32, 32, generator=g)
    ehs = torch.randn(2, 77, 768, generator=g) * 0.5
    with torch.no_grad():
        ref = unet3d_forward(w, cfg, x, 500, ehs)
    m = UNet3DConditionModel.from_config(unet_config("v1"))
    missing, unexpected = m.load_state_dict(w, strict=False)
    assert not missing and not unexpected
    del w
    m.to(DEV).prepare(DEV, torch.float16)
    from controlanimate_amd import kernels as K
    K._plan_sink = labels = []
    try:
        out = m(x.to(DEV), 500, ehs.to(DEV)).sample
    finally:
        K._plan_sink = None
    torch.cuda.synchronize()
    # 2 x 8 frames x 1024 tokens = 16384 rows: the one-launch forms of the C = 320 level run here too -- the temporal one in its
    # 8-frame form (two pixels per MFMA row tile)
    assert labels.count("tattn_out128") == 10 and labels.count("xattn_out128") == 5 and labels.count("ff_out128") >= 5, \
        {k: labels.count(k) for k in set(labels) if "out128" in k}
    r = rel(out, ref)
    print(f"full-width config-1 UNet3D eps rel_l2 = {r:.3e}")
    assert r < 1e-2, f"rel_l2 {r:.3e}"   # BASELINE north_star tolerance


def test_32_frames_full_width_unet_eps_vs_oracle():
    """BASELINE config 5's frame count (32 = the length of the v2 motion modules' positional table) at full width against the fp32
    oracle: the temporal attention of the C = 320 level runs in its 32-frame one-launch form (two MFMA row tiles per pixel, 2 x 2 score
    blocks; /root/reference/animatediff/models/motion_module.py:251-331)."""
    from controlanimate_amd import kernels as K
    from controlanimate_amd.configs import unet_config
    from controlanimate_amd.unet import UNet3DConditionModel
    from oracle.unet3d import UNet3DConfig, init_unet3d_weights, unet3d_forward
    cfg = UNet3DConfig.v2()
    w = init_unet3d_weights(cfg, seed=0)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(1, 4, 32, 32, 32, generator=g)
    ehs = torch.randn(1, 77, 768, generator=g) * 0.5
    with torch.no_grad():
        ref = unet3d_forward(w, cfg, x, 500, ehs)
    m = UNet3DConditionModel.from_config(unet_config("v2"))
    missing, unexpected = m.load_state_dict(w, strict=False)
    assert not missing and not unexpected
    del w
    m.to(DEV).prepare(DEV, torch.float16)
    K._plan_sink = labels = []
    try:
        out = m(x.to(DEV), 500, ehs.to(DEV)).sample
    finally:
        K._plan_sink = None
    torch.cuda.synchronize()
    assert labels.count("tattn_out128") == 10, {k: labels.count(k) for k in set(labels) if "attn" in k}
    r = rel(out, ref)
    print(f"full-width 32-frame UNet3D eps rel_l2 = {r:.3e}")
    assert r < 1e-2, f"rel_l2 {r:.3e}"   # BASELINE north_star tolerance


def test_ip_adapter_sites_full_width_eps_vs_oracle():
    """The IP-Adapter's UNet (modules/ip_adapter.py:95-127: IPAttnProcessor2_0 on the 16 attn2 sites; modules/attention_processor.py:433-477)
    at full width against the fp32 oracle, at a size where the five 64x64-latent sites run in their one-launch form (ABI v13: text
    attention + image-prompt attention + to_out + residual; 4 frames x 4096 tokens = 16384 rows)."""
    from controlanimate_amd import kernels as K
    from controlanimate_amd.attention_processor import AttnProcessor2_0, IPAttnProcessor2_0
    from controlanimate_amd.configs import unet_config
    from controlanimate_amd.unet import UNet3DConditionModel
    from oracle.unet3d import UNet3DConfig, init_unet3d_weights, unet3d_forward
    cfg = UNet3DConfig.v2()
    w = init_unet3d_weights(cfg, seed=0)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 4, 4, 64, 64, generator=g)
    ehs = torch.cat([torch.randn(1, 77, 768, generator=g) * 0.5, torch.randn(1, 4, 768, generator=g)], 1)
    m = UNet3DConditionModel.from_config(unet_config("v2"))
    missing, unexpected = m.load_state_dict(w, strict=False)
    assert not missing and not unexpected
    ip, procs = {}, {}
    for name in m.attn_processors.keys():
        if "attn2" not in name:
            procs[name] = AttnProcessor2_0()
            continue
        hidden = m.get_submodule(name[: -len(".processor")]).to_q.out_features
        p = IPAttnProcessor2_0(hidden_size=hidden, cross_attention_dim=768, scale=0.6, num_tokens=4)
        k = torch.randn(hidden, 768, generator=g) * 768 ** -0.5
        v = torch.randn(hidden, 768, generator=g) * 768 ** -0.5
        p.to_k_ip.weight.data.copy_(k)
        p.to_v_ip.weight.data.copy_(v)
        procs[name] = p
        ip[name[: -len(".processor")]] = {"to_k_ip": k, "to_v_ip": v, "scale": 0.6, "num_tokens": 4}
    assert len(ip) == 16
    with torch.no_grad():
        ref = unet3d_forward(w, cfg, x, 500, ehs, ip=ip)
        ref_text_only = unet3d_forward(w, cfg, x, 500, ehs[:, :77])
    del w
    m.set_attn_processor(procs)
    m.to(DEV).prepare(DEV, torch.float16)
    K._plan_sink = labels = []
    try:
        out = m(x.to(DEV), 500, ehs.to(DEV)).sample
    finally:
        K._plan_sink = None
    torch.cuda.synchronize()
    assert labels.count("xattn_ip_out128") == 5, {k: labels.count(k) for k in set(labels) if "attn" in k}
    r = rel(out, ref)
    print(f"full-width IP-Adapter UNet3D eps rel_l2 = {r:.3e} (the image-prompt tokens move eps by {rel(ref_text_only, ref):.3e})")
    assert r < 1e-2, f"rel_l2 {r:.3e}"      # BASELINE north_star tolerance
    assert rel(ref_text_only, ref) > 10 * r  # ... and the comparison would see the tokens missing


def test_config2_full_size_eps_vs_oracle():
    """/root/reference/animatediff/models/unet.py:458-621 and modules/controlresiduals_pipeline.py:278-316 at the benchmark
    workload: eps and all 13 ControlNet residuals within the north-star tolerance (1e-2 relative L2) of the fp32 oracle."""
    import gc
    import os
    from controlanimate_amd import kernels as K
    from controlanimate_amd.configs import controlnet_config, unet_config
    from controlanimate_amd.controlnet import ControlNetModel
    from controlanimate_amd.controlresiduals_pipeline import MultiControlNetResidualsPipeline
    from controlanimate_amd.unet import UNet3DConditionModel
    from oracle.controlnet import ControlNetConfig, init_controlnet_weights, multi_controlnet_residuals
    from oracle.unet3d import UNet3DConfig, init_unet3d_weights, unet3d_forward
    f, hw, t = 16, 64, 481
    ucfg, ccfg = UNet3DConfig.v2(), ControlNetConfig()
    uw, cw = init_unet3d_weights(ucfg, seed=0), init_controlnet_weights(ccfg, seed=1)
    g = torch.Generator().manual_seed(11)
    lat = torch.randn(1, 4, f, hw, hw, generator=g)
    pos, neg = torch.randn(1, 77, 768, generator=g) * 0.5, torch.randn(1, 77, 768, generator=g) * 0.5
    prompt = torch.cat([neg, pos])
    hints = torch.rand(f, 3, 8 * hw, 8 * hw, generator=g)
    x2 = torch.cat([lat] * 2)
    # ---- HIP first (then its models are freed: the oracle needs the host memory, not the device)
    unet = UNet3DConditionModel.from_config(unet_config("v2"))
    missing, unexpected = unet.load_state_dict(uw, strict=False)
    assert not missing and not unexpected
    unet.to(DEV).prepare(DEV, torch.float16)
    net = ControlNetModel.from_config(controlnet_config())
    net.load_state_dict(cw)
    net.to(DEV).prepare(DEV, torch.float16)
    cn = MultiControlNetResidualsPipeline(["c"], [1.0], use_lcm=False, controlnets=[net], device=DEV)
    cn.prep_control_images([h for h in hints], do_classifier_free_guidance=True, guess_mode=False)
    K._plan_sink = labels = []
    try:
        down, mid = cn(x2.to(DEV), t, prompt.to(DEV), f, do_classifier_free_guidance=True, guess_mode=False)
        out = unet(x2.to(DEV), t, prompt.to(DEV), down_block_additional_residuals=down, mid_block_additional_residual=mid).sample
        torch.cuda.synchronize()
    finally:
        K._plan_sink = None
    got = [d.float().cpu() for d in list(down) + [mid]] + [out.float().cpu()]
    kinds = sorted(set(labels))
    print("kernel labels of the step:", {k: labels.count(k) for k in kinds})
    # the step really ran the headline instantiations, not the small-shape ones
    assert any(k.startswith("wres") or k.startswith("ps") for k in kinds) and any("320" in k for k in kinds), kinds
    # ---- the PRODUCT's step, as ControlAnimationPipeline.model_eps runs it and bench.py times it: channels-last latents in,
    # both CFG halves declared identical (shared prefix), the ControlNet stack on the second stream with its 13 residual adds
    # inside the zero convolutions' epilogues, device-side timestep, the whole thing captured ONCE as a hipGraph and replayed.
    # Held to the same fp32 oracle, directly -- not through a chain of bit-equality tests -- and every one-launch /
    # headline kernel must have run: a `_supported` gate that says no at a config-2 shape FAILS here instead of falling back.
    lat_d, prompt_d = lat.to(DEV), prompt.to(DEV)
    x = K.latents_to_nhwc(lat_d, unet.conv_in.cin_pad, 2, 1.0, torch.float16)
    t_dev = torch.full((1,), float(t), device=DEV)

    def product_step():
        dn = cn.residuals_nhwc_async(x, t_dev, prompt_d, False, cfg_identical_halves=True, fuse_images=x.shape[0])
        return unet.forward_nhwc(x, 2, f, t_dev, prompt_d, dn, None, cfg_identical_halves=True)

    K._plan_sink = plabels = []
    try:
        eps_eager = product_step().clone()   # (also warms the per-window caches and the allocator before the capture)
        torch.cuda.synchronize()
    finally:
        K._plan_sink = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        eps_static = product_step()
    eps_static.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(eps_static, eps_eager), "hipGraph replay differs from the eager product step"
    eps_product = K.nhwc_to_ncfhw_f32(eps_static, 2, 4, f).cpu()
    pk = {k: plabels.count(k) for k in sorted(set(plabels))}
    print("kernel labels of the PRODUCT step:", pk)
    for need in ("ar128x64", "ff_out", "tattn_out", "xattn_out", "pq256x320", "attn_dma40", "attn_dma80", "attn_short", "wres160", "ps128x320"):
        assert any(k.startswith(need) for k in pk), f"the product step at config-2 size never ran `{need}`: {pk}"
    # ... and at every site of the 64x64-latent level, not just once: 7 text cross-attentions (UNet 5 + ControlNet 2), 10 temporal
    # attentions (5 motion modules x 2), 12 feed-forwards -- a `_supported` gate that declines SOME of them only costs time, silently
    for label, sites in (("xattn_out128", 7), ("tattn_out128", 10), ("ff_out128", 12)):
        assert pk.get(label, 0) >= sites, f"`{label}` ran {pk.get(label, 0)} times in the product step, expected {sites}: {pk}"
    del graph, eps_static, eps_eager
    del unet, net, cn, down, mid, out
    gc.collect()
    torch.cuda.empty_cache()
    # ---- the oracle, on every host core
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 32)))
    with torch.no_grad():
        down_o, mid_o = multi_controlnet_residuals([cw], ccfg, x2, t, prompt, f, [torch.cat([hints] * 2)], [1.0], guess_mode=False)
        ref = unet3d_forward(uw, ucfg, x2, t, prompt, down_o, mid_o)
    want = list(down_o) + [mid_o, ref]
    errs = [rel(a, b) for a, b in zip(got, want)]
    print("config-2 full size: residual rel_l2 max = %.3e, eps rel_l2 = %.3e" % (max(errs[:-1]), errs[-1]))
    assert len(errs) == 14
    for i, e in enumerate(errs[:-1]):
        assert e < 1e-2, f"ControlNet residual {i}: rel_l2 {e:.3e}"
    assert errs[-1] < 1e-2, f"eps rel_l2 {errs[-1]:.3e}"   # BASELINE north_star tolerance
    e_prod = rel(eps_product, ref)
    print("config-2 full size, PRODUCT path (forward_nhwc + shared CFG prefix + fused adds + second stream + hipGraph): eps rel_l2 = %.3e" % e_prod)
    assert e_prod < 1e-2, f"product-path eps rel_l2 {e_prod:.3e}"


def test_config2_size_properties():
    from controlanimate_amd import kernels as K
    from controlanimate_amd.configs import controlnet_config, unet_config
    from controlanimate_amd.controlnet import ControlNetModel
    from controlanimate_amd.controlresiduals_pipeline import MultiControlNetResidualsPipeline
    from controlanimate_amd.unet import UNet3DConditionModel
    torch.manual_seed(0)
    with torch.device(DEV):
        unet = UNet3DConditionModel.from_config(unet_config("v2"))
        net = ControlNetModel.from_config(controlnet_config())
    for mod in (unet, net):   # zero-initialised projections would hide the motion modules / ControlNet
        for p in mod.parameters():
            if p.dim() > 1 and float(p.detach().abs().max()) == 0.0:
                p.data.normal_(std=0.02)
    unet.prepare(DEV, torch.float16)
    net.prepare(DEV, torch.float16)
    f, hw = 16, 64
    g = torch.Generator().manual_seed(3)
    lat = torch.randn(1, 4, f, hw, hw, generator=g).to(DEV)
    pos = (torch.randn(1, 77, 768, generator=g) * 0.5).to(DEV)
    prompt_same = torch.cat([pos, pos]).contiguous()
    hints = torch.rand(f, 3, 512, 512, generator=g)
    cn = MultiControlNetResidualsPipeline(["c"], [1.0], use_lcm=True, controlnets=[net], device=DEV)  # b=1 residuals
    cn.prep_control_images([h for h in hints], do_classifier_free_guidance=True, guess_mode=False)
    x2 = K.latents_to_nhwc(lat, unet.conv_in.cin_pad, 2, 1.0, torch.float16)
    down, mid = cn.residuals_nhwc(x2[:f], 481, pos, False)
    eps_a = unet.forward_nhwc(x2, 2, f, 481, prompt_same, down, mid)
    eps_b = unet.forward_nhwc(x2, 2, f, 481, prompt_same, down, mid)
    torch.cuda.synchronize()
    assert torch.isfinite(eps_a).all()
    assert torch.equal(eps_a, eps_b), "not bit-deterministic"
    assert torch.equal(eps_a[:f], eps_a[f:]), "identical CFG halves must give identical eps"
    # b=1 residuals broadcast over the CFG batch == explicitly tiled residuals
    down2 = [torch.cat([d, d]) for d in down]
    eps_c = unet.forward_nhwc(x2, 2, f, 481, prompt_same, down2, torch.cat([mid, mid]))
    torch.cuda.synchronize()
    assert torch.equal(eps_a, eps_c)
    # the ControlNet really contributes
    eps_d = unet.forward_nhwc(x2, 2, f, 481, prompt_same)
    assert rel(eps_d, eps_a) > 1e-3


def test_config2_step_repeats_bit_identically_with_the_controlnet_stream_beside_the_unet():
    """tools/determinism_stress.py inside the suite: the full config-2 step as the product runs it -- ControlNet stack on the
    second HIP stream while the UNet encoder runs, residual adds in the zero convolutions' epilogues -- eight times; every
    eps (which the 13 residuals enter) equals the first, bit for bit.  (The streaming kernels confirm their global -> LDS
    units through LDS flags and the two streams compete for CUs: a race would show up as a differing repeat.)"""
    from controlanimate_amd import kernels as K
    from controlanimate_amd.configs import controlnet_config, unet_config
    from controlanimate_amd.controlnet import ControlNetModel
    from controlanimate_amd.controlresiduals_pipeline import MultiControlNetResidualsPipeline
    from controlanimate_amd.unet import UNet3DConditionModel
    torch.manual_seed(0)
    with torch.device(DEV):
        unet = UNet3DConditionModel.from_config(unet_config("v2"))
        net = ControlNetModel.from_config(controlnet_config())
    for mod in (unet, net):
        for p in mod.parameters():
            if p.dim() > 1 and float(p.detach().abs().max()) == 0.0:
                p.data.normal_(std=0.02)
    unet.prepare(DEV, torch.float16)
    net.prepare(DEV, torch.float16)
    f, hw = 16, 64
    g = torch.Generator().manual_seed(9)
    lat = torch.randn(1, 4, f, hw, hw, generator=g).to(DEV)
    prompt = (torch.randn(2, 77, 768, generator=g) * 0.5).to(DEV)
    cn = MultiControlNetResidualsPipeline(["c"], [1.0], use_lcm=False, controlnets=[net], device=DEV)  # (hints doubled: ControlNet batch 32)
    cn.prep_control_images([h for h in torch.rand(f, 3, 512, 512, generator=g)], do_classifier_free_guidance=True, guess_mode=False)
    x2 = K.latents_to_nhwc(lat, unet.conv_in.cin_pad, 2, 1.0, torch.float16)
    t = torch.full((1,), 481.0, device=DEV)

    def step():
        # (the calls of ControlAnimationPipeline's model_eps: both CFG halves of the latents are the same tensor)
        down = cn.residuals_nhwc_async(x2, t, prompt, False, cfg_identical_halves=True, fuse_images=x2.shape[0])
        eps = unet.forward_nhwc(x2, 2, f, t, prompt, down, None, cfg_identical_halves=True)
        torch.cuda.synchronize()
        return eps.clone()

    first = step()
    assert torch.isfinite(first).all()
    differing = sum(int(not torch.equal(step(), first)) for _ in range(8))
    assert differing == 0, f"{differing} of 8 repeats differ from the first"


def test_config3_step_four_full_width_controlnets_on_two_lanes_repeats_bit_identically():
    """BASELINE config 3 at full size, as the product runs it: four full-width ControlNets whose bodies run
    on two HIP streams beside the UNet encoder (16 frames, 64x64 latents, CFG batch 2), joined for the zero convolutions with the
    residual adds fused -- eight repeats eagerly and eight replays of ONE captured hipGraph, every eps equal to the first bit
    for bit, finite, the two CFG halves equal under identical conditioning, equal to the one-lane order, and the stack moves eps.
    (/root/reference/modules/controlresiduals_pipeline.py:278-316 with SampleConfig.yaml's four nets.)"""
    from controlanimate_amd import kernels as K
    from controlanimate_amd.configs import controlnet_config, unet_config
    from controlanimate_amd.context import dispatch
    from controlanimate_amd.controlnet import ControlNetModel
    from controlanimate_amd.controlresiduals_pipeline import MultiControlNetResidualsPipeline
    from controlanimate_amd.unet import UNet3DConditionModel
    torch.manual_seed(0)
    with torch.device(DEV):
        unet = UNet3DConditionModel.from_config(unet_config("v2"))
        nets = [ControlNetModel.from_config(controlnet_config()) for _ in range(4)]
    for mod in (unet, *nets):
        for p in mod.parameters():
            if p.dim() > 1 and float(p.detach().abs().max()) == 0.0:
                p.data.normal_(std=0.02)
    unet.prepare(DEV, torch.float16)
    for n in nets:
        n.prepare(DEV, torch.float16)
    f, hw = 16, 64
    g = torch.Generator().manual_seed(19)
    lat = torch.randn(1, 4, f, hw, hw, generator=g).to(DEV)
    p1 = (torch.randn(1, 77, 768, generator=g) * 0.5).to(DEV)
    prompt = torch.cat([p1, p1]).contiguous()   # both halves conditioned identically
    cn = MultiControlNetResidualsPipeline(list("abcd"), [1.0, 0.8, 0.6, 0.4], use_lcm=False, controlnets=nets, device=DEV)
    cn.prep_control_images({k: [h for h in torch.rand(f, 3, 512, 512, generator=g)] for k in "abcd"},
                           do_classifier_free_guidance=True, guess_mode=False)
    x2 = K.latents_to_nhwc(lat, unet.conv_in.cin_pad, 2, 1.0, torch.float16)
    t = torch.full((1,), 481.0, device=DEV)

    def enqueue():
        down = cn.residuals_nhwc_async(x2, t, prompt, False, cfg_identical_halves=True, fuse_images=x2.shape[0])
        return unet.forward_nhwc(x2, 2, f, t, prompt, down, None, cfg_identical_halves=True)

    def step():
        eps = enqueue()
        torch.cuda.synchronize()
        return eps.clone()

    assert dispatch.controlnet_streams == 2
    first = step()
    assert cn.lanes_used == 2
    assert torch.isfinite(first).all() and torch.equal(first[:f], first[f:])
    differing = sum(int(not torch.equal(step(), first)) for _ in range(8))
    assert differing == 0, f"{differing} of 8 eager repeats differ from the first"
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        eps_static = enqueue()
    bad = 0
    for _ in range(8):
        eps_static.zero_()
        graph.replay()
        torch.cuda.synchronize()
        bad += int(not torch.equal(eps_static, first))
    assert bad == 0, f"{bad} of 8 replays of the captured two-lane step differ from the eager step"
    del graph, eps_static
    dispatch.controlnet_streams = 1
    try:
        one_lane = step()
        assert cn.lanes_used == 1
    finally:
        dispatch.controlnet_streams = 2
    assert torch.equal(one_lane, first), "four bodies on two streams differ from four bodies on one"
    no_cn = unet.forward_nhwc(x2, 2, f, t, prompt, None, None, cfg_identical_halves=True)
    torch.cuda.synchronize()
    assert rel(no_cn, first) > 1e-3   # the four nets really contribute


def test_config1_shape_v2_full_width_winograd_paths_agree():
    """BASELINE config 1's shape (8 frames, 32x32 latents: levels 32, 16, 8, 4) on the full-width mm-v2 UNet3D: the 8x8- and 16x16-latent
    convolutions take the Winograd route (with the GroupNorm writing the transformed input where it can), the 4x4 level -- 64 tiles --
    must keep the direct split-K form.  The fused GroupNorm -> V form is bit-identical to GroupNorm + convolution; against the direct
    convolutions the route moves eps by fp16 rounding only."""
    from controlanimate_amd import kernels as K
    from controlanimate_amd.configs import unet_config
    from controlanimate_amd.context import dispatch
    from controlanimate_amd.unet import UNet3DConditionModel
    torch.manual_seed(0)
    with torch.device(DEV):
        unet = UNet3DConditionModel.from_config(unet_config("v2"))
    for p in unet.parameters():
        if p.dim() > 1 and float(p.detach().abs().max()) == 0.0:
            p.data.normal_(std=0.02)
    unet.prepare(DEV, torch.float16)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 4, 8, 32, 32, generator=g).to(DEV)
    ehs = (torch.randn(2, 77, 768, generator=g) * 0.5).to(DEV)
    K._plan_sink = labels = []
    try:
        fused = unet(x, 500, ehs).sample
    finally:
        K._plan_sink = None
    assert torch.isfinite(fused).all()
    assert any(l == "gn_wino_pq256x320" for l in labels) and any(l == "wino_pq256x320" for l in labels), sorted(set(labels))
    try:
        dispatch.gn_winograd = False
        separate = unet(x, 500, ehs).sample
        dispatch.conv_winograd = False
        direct = unet(x, 500, ehs).sample
    finally:
        dispatch.gn_winograd = dispatch.conv_winograd = True
    assert torch.equal(fused, separate)
    # (two fp16 paths whose rounding errors are independent: each sits 1.4e-3 .. 2.4e-3 from the fp32 oracle -- tools/bf16_fullwidth_gpu.py --
    #  so their distance is of that order; a wrong transform would be off by O(1))
    assert rel(fused, direct) < 5e-3
