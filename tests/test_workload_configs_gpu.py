"""GPU: BASELINE configs 3, 4 and 5 at their real workload.

The fp32 oracle needs minutes per forward at these sizes, so full-size runs are checked through size-independent
properties -- determinism, finiteness, equality of the two CFG halves under identical conditioning, equivalence of a
broadcast (b = 1) ControlNet residual with the explicitly doubled one -- while parity against the oracle runs at reduced
width (4 ControlNets summed) and at kernel level (spatial self-attention at the 6144- and 9216-token sequence lengths of
the 512x768 and 768x768 configurations, head dim 40, against a chunked fp32 reference)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SMALL = (64, 128, 256, 256)


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item()


def _full_unet(ip=False):
    from types import SimpleNamespace
    from controlanimate_amd.configs import unet_config
    from controlanimate_amd.unet import UNet3DConditionModel
    torch.manual_seed(0)
    with torch.device(DEV):
        unet = UNet3DConditionModel.from_config(unet_config("v2"))
    g = torch.Generator().manual_seed(1)
    for p in unet.parameters():  # zero-initialised projections (motion proj_out) would hide half of the network
        if p.dim() > 1 and float(p.detach().abs().max()) == 0.0:
            p.data.copy_((torch.randn(p.shape, generator=g) * 0.02).to(DEV))
    if ip:
        from controlanimate_amd.ip_adapter import IPAdapter
        ipa = IPAdapter(SimpleNamespace(unet=unet), None, None, DEV, num_tokens=4)
        for proc in unet.attn_processors.values():
            if hasattr(proc, "to_k_ip"):
                for lin in (proc.to_k_ip, proc.to_v_ip):
                    lin.weight.data.copy_((torch.randn(lin.weight.shape, generator=g) * lin.weight.shape[1] ** -0.5).to(DEV))
        ipa.set_scale(0.4)
    return unet.prepare(DEV, torch.float16)


def _full_controlnet(seed, strip_ip):
    from controlanimate_amd.attention_processor import CNAttnProcessor2_0
    from controlanimate_amd.configs import controlnet_config
    from controlanimate_amd.controlnet import ControlNetModel
    torch.manual_seed(seed)
    with torch.device(DEV):
        net = ControlNetModel.from_config(controlnet_config())
    g = torch.Generator().manual_seed(seed)
    for p in net.parameters():
        if p.dim() > 1 and float(p.detach().abs().max()) == 0.0:  # zero-convs: trained checkpoints are non-zero
            p.data.copy_((torch.randn(p.shape, generator=g) * 0.02).to(DEV))
    if strip_ip:
        net.set_attn_processor(CNAttnProcessor2_0(num_tokens=4))
    return net.prepare(DEV, torch.float16)


def test_config4_full_size_ip_adapter_two_guess_mode_controlnets():
    """512x768 (latents 64x96: N = 6144 / 1536 / 384 / 96), 16 frames, CFG batch 2, 81-token context, IP-Adapter on the
    UNet, 2 ControlNets in guess mode (b = 1, logspace scales, residuals broadcast over the CFG batch)."""
    from controlanimate_amd import kernels as K
    from controlanimate_amd.controlresiduals_pipeline import MultiControlNetResidualsPipeline
    f, h, w = 16, 64, 96
    unet = _full_unet(ip=True)
    nets = [_full_controlnet(10 + i, strip_ip=True) for i in range(2)]
    g = torch.Generator().manual_seed(5)
    lat = torch.randn(1, 4, f, h, w, generator=g).to(DEV)
    pos = torch.cat([torch.randn(1, 77, 768, generator=g) * 0.5, torch.randn(1, 4, 768, generator=g)], 1).to(DEV)
    prompt_same = torch.cat([pos, pos]).contiguous()          # both CFG halves conditioned identically
    hints = torch.rand(f, 3, 8 * h, 8 * w, generator=g)
    cn = MultiControlNetResidualsPipeline(["a", "b"], [1.0, 0.6], use_lcm=False, controlnets=nets, device=DEV)
    cn.prep_control_images([x for x in hints], do_classifier_free_guidance=True, guess_mode=True)
    assert cn.prep_images[0].shape[0] == f                      # guess mode: hints are NOT doubled (:268-269)
    cpad = unet.conv_in.cin_pad
    x2 = K.latents_to_nhwc(lat, cpad, 2, 1.0, torch.float16)    # [(2 f), h, w, 8]
    down, mid = cn.residuals_nhwc(x2[:f], 500, pos, True)       # b = 1 input (:811-813)
    assert len(down) == 12 and down[0].shape[0] == f and mid.shape[0] == f
    eps = unet.forward_nhwc(x2, 2, f, 500, prompt_same, down, mid)
    eps_again = unet.forward_nhwc(x2, 2, f, 500, prompt_same, down, mid)
    torch.cuda.synchronize()
    assert eps.shape == (2 * f, h, w, 4) and torch.isfinite(eps).all()
    def same(a, b, what):  # bit-for-bit, and say how far apart if not (a race shows up as a few scattered elements)
        if not torch.equal(a, b):
            d = (a.float() - b.float()).abs()
            raise AssertionError(f"{what}: {int((d > 0).sum())} of {d.numel()} elements differ, max |d| = {float(d.max()):.3e}, "
                                 f"first at flat index {int((d.flatten() > 0).nonzero()[0])}")
    same(eps, eps_again, "two identical forwards")              # deterministic kernels (no atomics)
    same(eps[:f], eps[f:], "CFG halves with identical conditioning")   # identical halves -> identical results, bit for bit
    assert 0.05 < float(eps.float().std()) < 50
    # broadcast residuals (b = 1) == the same residuals written out for both halves
    down2 = [torch.cat([d, d]) for d in down]
    eps_doubled = unet.forward_nhwc(x2, 2, f, 500, prompt_same, down2, torch.cat([mid, mid]))
    torch.cuda.synchronize()
    same(eps, eps_doubled, "broadcast vs written-out ControlNet residuals")
    # the image tokens matter on the UNet (IP branch) and are ignored by the ControlNets (CN processor)
    pos_other = pos.clone()
    pos_other[:, 77:] = torch.randn(1, 4, 768, generator=g).to(DEV)
    down_o, mid_o = cn.residuals_nhwc(x2[:f], 500, pos_other, True)
    for k, (a, b) in enumerate(zip(list(down) + [mid], list(down_o) + [mid_o])):
        same(a, b, f"ControlNet residual {k} with other image tokens")
    eps_o = unet.forward_nhwc(x2, 2, f, 500, torch.cat([pos_other, pos_other]).contiguous(), down, mid)
    assert not torch.equal(eps, eps_o)
    # the five IP-Adapter sites of the 64x96-latent level run as ONE launch each (ABI v13: text + image-prompt attention + to_out +
    # residual); the separate launches they replace give the same eps to fp16 rounding
    from controlanimate_amd.context import dispatch
    K._plan_sink = labels = []
    try:
        same(eps, unet.forward_nhwc(x2, 2, f, 500, prompt_same, down, mid), "a third identical forward")
    finally:
        K._plan_sink = None
    assert labels.count("xattn_ip_out128") == 5 and labels.count("xattn_out128") == 0, {k: labels.count(k) for k in set(labels) if "attn" in k}
    dispatch.xattn_ip_fused = False
    try:
        eps_sep = unet.forward_nhwc(x2, 2, f, 500, prompt_same, down, mid)
    finally:
        dispatch.xattn_ip_fused = True
    # (two fp16 evaluations of the whole network that round at different places: 2.2e-3 measured, the same distance the Winograd and the
    #  direct convolutions put between two runs in tests/test_fullsize_gpu.py; each is held to the oracle at the sizes the oracle reaches)
    assert rel(eps, eps_sep) < 5e-3, rel(eps, eps_sep)
    # a prompt tensor rewritten IN PLACE (what the pipeline does between windows): the cached text AND image-prompt K / V fragments
    # are re-projected and repacked at their addresses (refresh_window_caches)
    p2 = prompt_same.clone()
    same(eps, unet.forward_nhwc(x2, 2, f, 500, p2, down, mid), "the same prompt in another tensor")
    p2.copy_(torch.cat([pos_other, pos_other]))
    same(eps_o, unet.forward_nhwc(x2, 2, f, 500, p2, down, mid), "the prompt tensor rewritten in place")


def test_config5_full_size_32_frames_768():
    """768x768 (latents 96x96: N = 9216 / 2304 / 576 / 144), 32 frames (= the positional-encoding length), CFG batch 2:
    B = 64 images per step."""
    from controlanimate_amd import kernels as K
    f, h = 32, 96
    unet = _full_unet()
    g = torch.Generator().manual_seed(6)
    lat = torch.randn(1, 4, f, h, h, generator=g).to(DEV)
    pos = (torch.randn(1, 77, 768, generator=g) * 0.5).to(DEV)
    neg = (torch.randn(1, 77, 768, generator=g) * 0.5).to(DEV)
    x2 = K.latents_to_nhwc(lat, unet.conv_in.cin_pad, 2, 1.0, torch.float16)
    same = torch.cat([pos, pos]).contiguous()
    eps = unet.forward_nhwc(x2, 2, f, 999, same)
    eps2 = unet.forward_nhwc(x2, 2, f, 999, same)
    torch.cuda.synchronize()
    assert eps.shape == (2 * f, h, h, 4) and torch.isfinite(eps).all()
    assert torch.equal(eps, eps2) and torch.equal(eps[:f], eps[f:])
    diff = unet.forward_nhwc(x2, 2, f, 999, torch.cat([neg, pos]).contiguous())
    assert torch.equal(diff[f:], eps[f:]) and not torch.equal(diff[:f], eps[:f])   # the halves are independent problems
    # temporal coupling exists: changing ONE frame's latents changes the eps of the other frames
    lat_b = lat.clone()
    lat_b[:, :, 7] += 1.0
    xb = K.latents_to_nhwc(lat_b, unet.conv_in.cin_pad, 2, 1.0, torch.float16)
    eps_b = unet.forward_nhwc(xb, 2, f, 999, same)
    assert not torch.equal(eps_b[0], eps[0])


def test_config3_four_controlnets_summed_vs_oracle():
    """SampleConfig-equivalent stack of 4 ControlNets with the reference's non-guess CFG behaviour (hints doubled, the
    [neg,pos,neg,pos,...] prompt tiling quirk): summed residuals and UNet eps against the fp32 oracle at reduced width."""
    from controlanimate_amd.configs import controlnet_config, unet_config
    from controlanimate_amd.controlnet import ControlNetModel
    from controlanimate_amd.controlresiduals_pipeline import MultiControlNetResidualsPipeline
    from controlanimate_amd.unet import UNet3DConditionModel
    from oracle.controlnet import ControlNetConfig, init_controlnet_weights, multi_controlnet_residuals
    from oracle.unet3d import UNet3DConfig, init_unet3d_weights, unet3d_forward
    f, h = 8, 8
    ucfg, ccfg = UNet3DConfig.v2(block_out_channels=SMALL), ControlNetConfig(block_out_channels=SMALL)
    uw = init_unet3d_weights(ucfg, seed=31)
    unet = UNet3DConditionModel.from_config(unet_config("v2", block_out_channels=SMALL))
    unet.load_state_dict(uw)
    unet.to(DEV).prepare(DEV, torch.float16)
    cws, nets = [], []
    for i in range(4):
        cw = init_controlnet_weights(ccfg, seed=40 + i)
        net = ControlNetModel.from_config(controlnet_config(block_out_channels=SMALL))
        net.load_state_dict(cw)
        nets.append(net.to(DEV).prepare(DEV, torch.float16))
        cws.append(cw)
    g = torch.Generator().manual_seed(32)
    lat = torch.randn(1, 4, f, h, h, generator=g)
    pos, neg = torch.randn(1, 77, 768, generator=g) * 0.5, torch.randn(1, 77, 768, generator=g) * 0.5
    prompt = torch.cat([neg, pos])
    hints = [torch.rand(f, 3, 8 * h, 8 * h, generator=g) for _ in range(4)]
    scales = [1.0, 0.8, 0.6, 0.4]
    x2 = torch.cat([lat] * 2)
    with torch.no_grad():
        down_o, mid_o = multi_controlnet_residuals(cws, ccfg, x2, 700, prompt, f, [torch.cat([hh] * 2) for hh in hints], scales, guess_mode=False)
        ref = unet3d_forward(uw, ucfg, x2, 700, prompt, down_o, mid_o)
    cn = MultiControlNetResidualsPipeline(list("abcd"), scales, use_lcm=False, controlnets=nets, device=DEV)
    cn.prep_control_images({k: [x for x in hh] for k, hh in zip("abcd", hints)}, do_classifier_free_guidance=True, guess_mode=False)
    down, mid = cn(x2.to(DEV), 700, prompt.to(DEV), f, do_classifier_free_guidance=True, guess_mode=False)
    assert len(down) == 12
    for i, (a, b) in enumerate(zip(list(down) + [mid], list(down_o) + [mid_o])):
        assert rel(a, b) < 1e-2, (i, rel(a, b))
    out = unet(x2.to(DEV), 700, prompt.to(DEV), down_block_additional_residuals=down, mid_block_additional_residual=mid).sample
    torch.cuda.synchronize()
    assert rel(out, ref) < 1e-2, rel(out, ref)


@pytest.mark.parametrize("tokens", [6144, 9216])
def test_spatial_attention_long_sequences_head_dim_40(tokens):
    """The level-0 self-attention of configs 4 / 5 (N = 6144 / 9216, 8 heads of 40) against fp32, chunked over queries."""
    from controlanimate_amd import kernels as K
    images, heads, d = 2, 8, 40
    c = heads * d
    g = torch.Generator().manual_seed(tokens)
    qkv = (torch.randn(images * tokens, 3 * c, generator=g) * 0.7).to(DEV).half()
    out = K.attention_spatial(qkv, images, tokens, heads)
    torch.cuda.synchronize()
    q, k, v = (qkv.float().reshape(images, tokens, 3, heads, d).permute(2, 0, 3, 1, 4))  # [images, heads, tokens, d] each
    ref = torch.empty(images, heads, tokens, d, device=DEV)
    for s in range(0, tokens, 1024):
        p = torch.softmax(q[:, :, s:s + 1024] @ k.transpose(-1, -2) * d ** -0.5, dim=-1)
        ref[:, :, s:s + 1024] = p @ v
    ref = ref.permute(0, 2, 1, 3).reshape(images * tokens, c)
    assert rel(out, ref) < 2e-3, rel(out, ref)


def test_ip_adapter_image_tokens_match_the_reference_projection():
    """ImageProjModel.forward and get_image_embeds_4controlanimate on HIP (modules/ip_adapter.py:30-47, 187-222) against
    the reference's own module (tests/golden/make_ip_golden.py): cond tokens = proj(embeds), uncond = proj(zeros)."""
    import os
    from types import SimpleNamespace
    import numpy as np
    from controlanimate_amd.ip_adapter import IPAdapter, ImageProjModel
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ip_image_proj.npz"))
    m = ImageProjModel(cross_attention_dim=768, clip_embeddings_dim=1024, clip_extra_context_tokens=4)
    g = torch.Generator().manual_seed(int(fx["weight_seed"]))   # the generator sequence of make_ip_golden.py
    sd = {"proj.weight": torch.randn(4 * 768, 1024, generator=g) * 1024 ** -0.5, "proj.bias": torch.randn(4 * 768, generator=g) * 0.1,
          "norm.weight": 1 + 0.1 * torch.randn(768, generator=g), "norm.bias": 0.1 * torch.randn(768, generator=g)}
    assert abs(float(sum(v.double().abs().sum() for v in sd.values())) - float(fx["weight_checksum"])) < 1e-6 * float(fx["weight_checksum"])
    m.load_state_dict(sd)
    m.to(DEV).prepare(DEV, torch.float16)
    emb = torch.from_numpy(fx["clip_image_embeds"]).to(DEV)
    tok = m(emb)
    assert tuple(tok.shape) == (2, 4, 768) and rel(tok, torch.from_numpy(fx["tokens"])) < 3e-3
    # the adapter's plumbing around it: scale is written into the IP processors, uncond tokens come from a zero embedding
    class _Proc:
        scale = 1.0
    ip = object.__new__(IPAdapter)
    ip.device, ip.image_encoder, ip.image_proj_model = torch.device(DEV), None, m
    from controlanimate_amd.attention_processor import IPAttnProcessor
    proc = IPAttnProcessor(hidden_size=64, cross_attention_dim=768, scale=1.0, num_tokens=4)
    ip.pipe = SimpleNamespace(unet=SimpleNamespace(attn_processors={"a": proc, "b": object()}))
    cond, uncond = ip.get_image_embeds_4controlanimate(clip_image_embeds=emb[:1], scale=0.35)
    assert proc.scale == 0.35
    assert tuple(cond.shape) == (1, 4, 768) and rel(cond, torch.from_numpy(fx["tokens"][:1])) < 3e-3
    assert rel(uncond, torch.from_numpy(fx["uncond"][:1])) < 3e-3


def test_hint_embedding_of_cfg_doubled_hints_is_computed_once():
    """`prep_control_images` doubles the control frames for classifier-free guidance (reference :268-269); the two halves
    are the same images, so the eight-convolution hint embedding runs on one half and is repeated: the same bits as
    embedding the doubled batch, half of the most expensive per-window work."""
    from controlanimate_amd.controlresiduals_pipeline import MultiControlNetResidualsPipeline
    net = _full_controlnet(21, strip_ip=False)
    g = torch.Generator().manual_seed(9)
    frames = torch.rand(4, 3, 128, 192, generator=g)
    cn = MultiControlNetResidualsPipeline(["a"], [1.0], use_lcm=False, controlnets=[net], device=DEV)
    cn.prep_control_images([x for x in frames], do_classifier_free_guidance=True, guess_mode=False)
    doubled = cn.prep_images[0]
    assert doubled.shape[0] == 8 and getattr(doubled, "_cfg_doubled", False)
    once = net.hint_embedding(doubled, DEV).clone()
    plain = doubled.clone()                       # the same values without the marker: embedded as 8 independent images
    assert not getattr(plain, "_cfg_doubled", False)
    full = net.hint_embedding(plain, DEV)
    torch.cuda.synchronize()
    assert once.shape == full.shape == (8, 16, 24, 320) and torch.equal(once, full) and torch.equal(once[:4], once[4:])
    cn.prep_control_images([x for x in frames], do_classifier_free_guidance=True, guess_mode=True)
    assert cn.prep_images[0].shape[0] == 4 and not getattr(cn.prep_images[0], "_cfg_doubled", False)


def test_shared_prefix_of_the_two_cfg_halves_changes_nothing():
    """Classifier-free guidance feeds the SAME latents to both batch halves (reference :797): conv_in, the first resnet and
    the first transformer's GroupNorm / proj_in / self-attention see identical data twice.  `cfg_identical_halves=True` runs
    them once and repeats the activations; eps and the ControlNet residuals must be what the full-batch path computes."""
    from controlanimate_amd import kernels as K
    from controlanimate_amd.controlresiduals_pipeline import MultiControlNetResidualsPipeline
    f, h, w = 4, 64, 64
    unet = _full_unet()
    net = _full_controlnet(31, strip_ip=False)
    g = torch.Generator().manual_seed(11)
    lat = torch.randn(1, 4, f, h, w, generator=g).to(DEV)
    neg, pos = ((torch.randn(1, 77, 768, generator=g) * 0.5).to(DEV) for _ in range(2))
    prompt = torch.cat([neg, pos]).contiguous()
    x2 = K.latents_to_nhwc(lat, unet.conv_in.cin_pad, 2, 1.0, torch.float16)
    cn = MultiControlNetResidualsPipeline(["a"], [0.8], use_lcm=False, controlnets=[net], device=DEV)
    cn.prep_control_images([x for x in torch.rand(f, 3, 8 * h, 8 * w, generator=g)], do_classifier_free_guidance=True, guess_mode=False)
    from controlanimate_amd.context import dispatch
    d_full, m_full = cn.residuals_nhwc(x2, 500, prompt, False)
    d_full, m_full = [d.clone() for d in d_full], m_full.clone()
    dispatch.cn_cfg_dedup = False   # (the ControlNet's shared PREFIX: what runs when its two halves are not the same problem end to end)
    try:
        d_sh, m_sh = cn.residuals_nhwc(x2, 500, prompt, False, cfg_identical_halves=True)
    finally:
        dispatch.cn_cfg_dedup = True
    for k, (a, b) in enumerate(zip(d_full + [m_full], list(d_sh) + [m_sh])):
        assert torch.equal(a, b), f"ControlNet residual {k}"
    eps_full = unet.forward_nhwc(x2, 2, f, 500, prompt, d_full, m_full)
    eps_sh = unet.forward_nhwc(x2, 2, f, 500, prompt, d_full, m_full, cfg_identical_halves=True)
    torch.cuda.synchronize()
    assert torch.isfinite(eps_sh).all() and torch.equal(eps_full, eps_sh)
    assert not torch.equal(eps_sh[:f], eps_sh[f:])   # (the halves do differ: different prompts)


def test_controlnet_cfg_halves_are_one_problem_and_run_once():
    """Non-guess classifier-free guidance: the reference tiles the ControlNet's prompt as torch.cat([embeds] * frame_count)
    (/root/reference/modules/controlresiduals_pipeline.py:292: image z of the (b f) batch reads embeds[z % 2]), feeds both halves
    the same latents (:797 of the pipeline) and the same control frames (:268-269) -- so with an even frame count image z and
    image z + f are the same problem.  `dispatch.cn_cfg_dedup` solves it once and writes the residuals for both halves.  Checked
    here at full width against the all-images run: both halves of the FULL run are bit-equal (the premise), the de-duplicated
    residuals equal them to fp16 rounding (another batch size may pick another tile plan), its two halves are bit-equal, the
    fused form adds each half of the UNet's skips to its own half, and an odd frame count (the premise fails) runs all images."""
    from controlanimate_amd import kernels as K
    from controlanimate_amd.context import dispatch
    from controlanimate_amd.controlresiduals_pipeline import MultiControlNetResidualsPipeline
    h, w = 32, 32
    net = _full_controlnet(41, strip_ip=False)
    g = torch.Generator().manual_seed(13)
    neg, pos = ((torch.randn(1, 77, 768, generator=g) * 0.5).to(DEV) for _ in range(2))
    prompt = torch.cat([neg, pos]).contiguous()
    for f, expect_twice in ((6, True), (5, False)):
        lat = torch.randn(1, 4, f, h, w, generator=g).to(DEV)
        x2 = K.latents_to_nhwc(lat, net.conv_in.cin_pad, 2, 1.0, torch.float16)
        cn = MultiControlNetResidualsPipeline(["a"], [0.8], use_lcm=False, controlnets=[net], device=DEV)
        cn.prep_control_images([x for x in torch.rand(f, 3, 8 * h, 8 * w, generator=g)], do_classifier_free_guidance=True, guess_mode=False)
        body = net.forward_body(x2, 500, prompt, cn.prep_images[0], 0.8, False, cfg_identical_halves=True)
        assert body[3] is expect_twice and body[1].shape[0] == (f if expect_twice else 2 * f)
        dispatch.cn_cfg_dedup = False
        try:
            d_all, m_all = cn.residuals_nhwc(x2, 500, prompt, False, cfg_identical_halves=True)
            d_all, m_all = [d.clone() for d in d_all], m_all.clone()
        finally:
            dispatch.cn_cfg_dedup = True
        d_one, m_one = cn.residuals_nhwc(x2, 500, prompt, False, cfg_identical_halves=True)
        torch.cuda.synchronize()
        for k, (a, b) in enumerate(zip(d_all + [m_all], list(d_one) + [m_one])):
            assert a.shape == b.shape and torch.isfinite(b).all()
            if f % 2 == 0:
                assert torch.equal(a[:f], a[f:]), f"residual {k}: the halves of the all-images run differ -- the premise is wrong"
                assert torch.equal(b[:f], b[f:])
            assert rel(b, a) < 3e-3, (k, rel(b, a))
        if expect_twice:  # fused: out = zero_conv(half) + skips, each half of the skips to its own half of the result
            outs, xm, scales, twice = body
            skips = [(torch.randn(2 * o.shape[0], *o.shape[1:], generator=g) * 0.5).half().to(DEV) for o in outs]
            mid_x = (torch.randn(2 * xm.shape[0], *xm.shape[1:], generator=g) * 0.5).half().to(DEV)
            fd, fm = net.apply_zero_convs(outs, xm, scales, (skips, mid_x), twice)
            pd, pm = net.apply_zero_convs(outs, xm, scales, None, twice)
            torch.cuda.synchronize()
            for a, s_, r in zip((*fd, fm), (*skips, mid_x), (*pd, pm)):
                assert torch.equal(a, K.add_bcast(s_, r))   # (one net: the epilogue's add rounds like the separate add, test_graph_gpu.py)
