"""GPU parity of the HIP UNet3D / modules against fixtures produced by the REFERENCE's own modules
(tests/golden, see make_golden.py) and against the fp32 oracle.  Everything runs through the C ABI.

Tolerance: BASELINE north_star -- 1e-2 relative on UNet eps (fp16-class tolerance); reported as
relative L2 error of the whole tensor.  fp16 activations are also checked with a tighter bound.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda:0"
SMALL = (64, 128, 256, 256)
# fp16 (the shipped default, = the reference's .half()) must meet the north_star bound 1e-2 and
# measures ~2.5e-3.  bf16: rounding ONLY the matrix-multiply operands of the exact fp32 oracle to bf16 already gives
# 1.39e-2 on this fixture (tests/test_bf16_floor_cpu.py) -- no bf16-operand implementation can meet 1e-2 here; the
# HIP path measures ~1.8e-2 and is held to 2.5e-2 (< 2x that floor).  bf16 is an opt-in range-safe mode.
TOL = {torch.bfloat16: 2.5e-2, torch.float16: 1e-2}


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def T(a):
    return torch.from_numpy(np.asarray(a))


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item()


def build_unet(version, boc, weights, dtype, **over):
    from controlanimate_amd.configs import unet_config
    from controlanimate_amd.unet import UNet3DConditionModel
    m = UNet3DConditionModel.from_config(unet_config(version, block_out_channels=boc, **over))
    missing, unexpected = m.load_state_dict(weights, strict=False)
    assert not unexpected and not missing, (missing[:3], unexpected[:3])
    return m.to(DEV).prepare(DEV, dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_unet3d_v2_against_reference_fixture(dtype):
    from oracle.unet3d import UNet3DConfig, init_unet3d_weights
    fx = load("unet3d_v2_w64.npz")
    w = init_unet3d_weights(UNet3DConfig.v2(block_out_channels=SMALL), seed=int(fx["weight_seed"]))
    m = build_unet("v2", SMALL, w, dtype)
    out = m(T(fx["sample"]).to(DEV), int(fx["timestep"]), T(fx["ehs"]).to(DEV)).sample
    torch.cuda.synchronize()
    assert out.shape == fx["out"].shape and out.dtype == torch.float32
    r = rel(out, T(fx["out"]))
    assert r < TOL[dtype], f"rel_l2 {r:.3e}"
    # second call with the same prompt tensor exercises the K/V cache; must be bit-identical
    ehs = T(fx["ehs"]).to(DEV)
    o1 = m(T(fx["sample"]).to(DEV), 500, ehs).sample
    o2 = m(T(fx["sample"]).to(DEV), 500, ehs).sample
    torch.cuda.synchronize()
    assert torch.equal(o1, o2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_unet3d_v1_cross_frame_gn_and_residual_broadcast(dtype):
    from oracle.unet3d import UNet3DConfig, init_unet3d_weights
    fx = load("unet3d_v1_w64.npz")
    w = init_unet3d_weights(UNet3DConfig.v1(block_out_channels=SMALL), seed=int(fx["weight_seed"]))
    m = build_unet("v1", SMALL, w, dtype)
    down = tuple(T(fx[f"down{i}"]).to(DEV) for i in range(12))
    out = m(T(fx["sample"]).to(DEV), torch.tensor(int(fx["timestep"])), T(fx["ehs"]).to(DEV),
            down_block_additional_residuals=down, mid_block_additional_residual=T(fx["mid"]).to(DEV), return_dict=False)[0]
    torch.cuda.synchronize()
    r = rel(out, T(fx["out"]))
    assert r < TOL[dtype], f"rel_l2 {r:.3e}"


def test_unet3d_native_lcm_timestep_cond():
    from controlanimate_amd.schedulers import get_w_embedding
    from oracle.unet3d import UNet3DConfig, init_unet3d_weights
    fx = load("unet3d_lcm_w64.npz")
    w = init_unet3d_weights(UNet3DConfig.v2(block_out_channels=SMALL, time_cond_proj_dim=256), seed=int(fx["weight_seed"]))
    m = build_unet("v2", SMALL, w, torch.float16, time_cond_proj_dim=256)
    wemb = get_w_embedding(torch.tensor([7.5]), embedding_dim=256)
    # sin/cos of fp32 arguments up to 7500 rad: the argument's own rounding (7500 * 2^-24) differs by
    # CPU vector-math library, hence the loose bound across machines
    assert torch.allclose(wemb, T(fx["w_embedding"]), atol=2e-3)
    out = m(T(fx["sample"]).to(DEV), torch.full((1,), int(fx["timestep"]), dtype=torch.long), T(fx["ehs"]).to(DEV),
            timestep_cond=wemb.to(DEV)).sample
    torch.cuda.synchronize()
    r = rel(out, T(fx["out"]))
    assert r < 1e-2, f"rel_l2 {r:.3e}"


def test_unet3d_ip_adapter_processors():
    from controlanimate_amd.attention_processor import AttnProcessor2_0, IPAttnProcessor2_0
    from oracle.unet3d import UNet3DConfig, init_unet3d_weights
    fx = load("unet3d_ip_w64.npz")
    cfg = UNet3DConfig.v2(block_out_channels=SMALL)
    w = init_unet3d_weights(cfg, seed=int(fx["weight_seed"]))
    m = build_unet("v2", SMALL, w, torch.float16)
    g = torch.Generator().manual_seed(int(fx["ip_seed"]))
    sites = [str(s) for s in fx["ip_sites"]]
    procs = {}
    for name in m.attn_processors.keys():
        if name not in sites:
            procs[name] = AttnProcessor2_0()
    for name in sites:  # weights were drawn in this order
        hidden = m.get_submodule(name[: -len(".processor")]).to_q.out_features
        p = IPAttnProcessor2_0(hidden_size=hidden, cross_attention_dim=768, scale=float(fx["ip_scale"]), num_tokens=4)
        p.to_k_ip.weight.data.copy_(torch.randn(hidden, 768, generator=g) * 768 ** -0.5)
        p.to_v_ip.weight.data.copy_(torch.randn(hidden, 768, generator=g) * 768 ** -0.5)
        procs[name] = p
    assert [k for k in m.attn_processors if "attn2" in k] == sites
    m.set_attn_processor(procs)
    m.prepare(DEV, torch.float16)
    out = m(T(fx["sample"]).to(DEV), int(fx["timestep"]), T(fx["ehs"]).to(DEV)).sample
    torch.cuda.synchronize()
    r = rel(out, T(fx["out"]))
    assert r < 1e-2, f"rel_l2 {r:.3e}"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fullwidth_modules_head_dims_40_80_160(dtype):
    """ResnetBlock3D / Transformer3DModel / VanillaTemporalModule at real SD1.5 widths."""
    from controlanimate_amd import kernels as K
    from controlanimate_amd.attention import Transformer3DModel
    from controlanimate_amd.configs import INFERENCE_V2
    from controlanimate_amd.context import ExecCtx
    from controlanimate_amd.layers import WeightArena
    from controlanimate_amd.motion_module import VanillaTemporalModule
    from controlanimate_amd.resnet import ResnetBlock3D
    from oracle.unet3d import UNet3DConfig, init_from_shapes, unet3d_param_shapes
    fx = load("modules_fullwidth.npz")
    shapes = unet3d_param_shapes(UNet3DConfig.v2())

    def weights(pre, seed):
        return init_from_shapes({k[len(pre) + 1:]: v for k, v in shapes.items() if k.startswith(pre + ".")}, seed=seed)

    def nhwc(x5):
        return K.ncfhw_to_nhwc(x5.to(DEV), x5.shape[1], dtype)

    def back(y, b, c, f):
        return K.nhwc_to_ncfhw_f32(y, b, c, f)

    tol = 8e-3 if dtype == torch.bfloat16 else 1.5e-3
    # ---- resnet 320 -> 640 with shortcut; per-frame and cross-frame GroupNorm
    w = weights("down_blocks.1.resnets.0", int(fx["resnet_seed"]))
    x5, temb = T(fx["resnet_x"]), T(fx["resnet_temb"])
    b, c, f, h, wd = x5.shape
    for infl in (1, 0):
        r = ResnetBlock3D(in_channels=320, out_channels=640, temb_channels=1280, eps=1e-5, groups=32, use_inflated_groupnorm=bool(infl))
        r.load_state_dict(w)
        arena = WeightArena()
        r.pack(arena, dtype)
        r.time_emb_proj.pack(arena, dtype)
        arena.finalize(DEV)
        r.temb_slice = (0, 640)
        tproj = r.time_emb_proj.run(K.silu_f32(temb.to(DEV)).to(dtype), out_f32=True)
        ctx = ExecCtx(b=b, f=f, dtype=dtype, temb=tproj, emb_groups=b, gn_frames_per_stat=1 if infl else f)
        out = back(r(nhwc(x5), ctx), b, 640, f)
        torch.cuda.synchronize()
        rr = rel(out, T(fx[f"resnet_out_inflated{infl}"]))
        assert rr < tol, f"resnet inflated={infl}: rel_l2 {rr:.3e}"
    # ---- spatial transformers (head dims 40 / 80 / 160)
    for name, cc, pre in (("tx320", 320, "down_blocks.0.attentions.0"), ("tx640", 640, "down_blocks.1.attentions.0"),
                          ("tx1280", 1280, "down_blocks.2.attentions.0")):
        t = Transformer3DModel(8, cc // 8, in_channels=cc, num_layers=1, cross_attention_dim=768, norm_num_groups=32)
        missing, unexpected = t.load_state_dict(weights(pre, int(fx[f"{name}_seed"])), strict=False)
        assert not missing and not unexpected
        arena = WeightArena()
        t.pack(arena, dtype)
        arena.finalize(DEV)
        x5 = T(fx[f"{name}_x"])
        b, c, f, h, wd = x5.shape
        ctx = ExecCtx(b=b, f=f, dtype=dtype, temb=None, emb_groups=b, ehs=T(fx[f"{name}_ehs"]).to(DEV, dtype), frames_per_kv=f)
        out = back(t(nhwc(x5), ctx), b, cc, f)
        torch.cuda.synchronize()
        rr = rel(out, T(fx[f"{name}_out"]))
        assert rr < tol, f"{name}: rel_l2 {rr:.3e}"
    # ---- motion modules
    for name, cc, pre in (("mm320", 320, "down_blocks.0.motion_modules.0"), ("mm640", 640, "down_blocks.1.motion_modules.0"),
                          ("mm1280", 1280, "down_blocks.2.motion_modules.0")):
        mm = VanillaTemporalModule(in_channels=cc, **INFERENCE_V2["motion_module_kwargs"])
        mm.load_state_dict(weights(pre, int(fx[f"{name}_seed"])))
        arena = WeightArena()
        mm.pack(arena, dtype)
        arena.finalize(DEV)
        x5 = T(fx[f"{name}_x"])
        b, c, f, h, wd = x5.shape
        ctx = ExecCtx(b=b, f=f, dtype=dtype, temb=None, emb_groups=b)
        out = back(mm(nhwc(x5), ctx), b, cc, f)
        torch.cuda.synchronize()
        rr = rel(out, T(fx[f"{name}_out"]))
        assert rr < tol, f"{name}: rel_l2 {rr:.3e}"


def test_missing_extension_fails_loudly(monkeypatch):
    """No silent fallback: if the .so is absent the product path raises."""
    from controlanimate_amd import _capi
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", "/nonexistent/libcontrolanimate_hip.so")
    with pytest.raises(_capi.CAHipUnavailable):
        _capi.lib()


def test_non_square_latents_12_frames_two_controlnets_vs_oracle():
    """BASELINE config 4 in miniature: non-square latents (16 x 24 -> token counts 384 / 96 / 24 / 6, none a
    multiple of the 128-query / 64-key tiles), 12 frames (not a power of two), 81-token context (77 + 4 IP tokens,
    stripped by the ControlNet processor), two ControlNets summed -- UNet eps against the fp32 oracle."""
    from controlanimate_amd.configs import controlnet_config
    from controlanimate_amd.controlnet import ControlNetModel
    from controlanimate_amd.controlresiduals_pipeline import MultiControlNetResidualsPipeline
    from oracle.controlnet import ControlNetConfig, controlnet_forward, init_controlnet_weights
    from oracle.unet3d import UNet3DConfig, init_unet3d_weights, unet3d_forward
    ucfg = UNet3DConfig.v2(block_out_channels=SMALL)
    uw = init_unet3d_weights(ucfg, seed=71)
    unet = build_unet("v2", SMALL, uw, torch.float16)
    ccfg = ControlNetConfig(block_out_channels=SMALL)
    f, h, w = 12, 16, 24
    g = torch.Generator().manual_seed(72)
    sample = torch.randn(1, 4, f, h, w, generator=g)
    ehs = torch.cat([torch.randn(1, 77, 768, generator=g) * 0.5, torch.randn(1, 4, 768, generator=g)], dim=1)  # 77 text + 4 image tokens
    assert ehs.shape[1] == 81
    hints = [torch.rand(f, 3, 8 * h, 8 * w, generator=g) for _ in range(2)]
    scales = [0.7, 1.1]
    # IP-Adapter processors on the UNet's 16 cross-attention sites (modules/ip_adapter.py:95-127) ...
    from controlanimate_amd.attention_processor import AttnProcessor2_0, CNAttnProcessor2_0, IPAttnProcessor2_0
    procs, ip_oracle = {}, {}
    for name in unet.attn_processors.keys():
        if "attn2" in name and "temporal_transformer" not in name:
            hidden = unet.get_submodule(name[: -len(".processor")]).to_q.out_features
            pr = IPAttnProcessor2_0(hidden_size=hidden, cross_attention_dim=768, scale=0.4, num_tokens=4)
            pr.to_k_ip.weight.data.copy_(torch.randn(hidden, 768, generator=g) * 768 ** -0.5)
            pr.to_v_ip.weight.data.copy_(torch.randn(hidden, 768, generator=g) * 768 ** -0.5)
            procs[name] = pr
            ip_oracle[name[: -len(".processor")]] = dict(to_k_ip=pr.to_k_ip.weight.data.clone(), to_v_ip=pr.to_v_ip.weight.data.clone(),
                                                         scale=0.4, num_tokens=4)
        else:
            procs[name] = AttnProcessor2_0()
    unet.set_attn_processor(procs)
    unet.prepare(DEV, torch.float16)
    nets, cws = [], []
    for i in range(2):
        cw = init_controlnet_weights(ccfg, seed=73 + i)
        net = ControlNetModel.from_config(controlnet_config(block_out_channels=SMALL))
        net.load_state_dict(cw)
        net.set_attn_processor(CNAttnProcessor2_0(num_tokens=4))  # ... and the token-stripping one on the ControlNets (:129-134)
        nets.append(net.to(DEV).prepare(DEV, torch.float16))
        cws.append(cw)
    # oracle: per-frame ControlNets on the (b f) batch, residuals summed, then the UNet
    with torch.no_grad():
        x2d = sample.permute(0, 2, 1, 3, 4).reshape(f, 4, h, w)
        down_sum, mid_sum = None, None
        for cw, hint, sc in zip(cws, hints, scales):
            d, m = controlnet_forward(cw, ccfg, x2d, 300, ehs.expand(f, -1, -1), hint, conditioning_scale=sc, guess_mode=False, strip_tokens=4)
            down_sum = list(d) if down_sum is None else [a + b for a, b in zip(down_sum, d)]
            mid_sum = m if mid_sum is None else mid_sum + m
        to5 = lambda t: t.reshape(1, f, *t.shape[1:]).permute(0, 2, 1, 3, 4)
        ref = unet3d_forward(uw, ucfg, sample, 300, ehs, down_block_additional_residuals=[to5(d) for d in down_sum],
                             mid_block_additional_residual=to5(mid_sum), ip=ip_oracle)
    cn = MultiControlNetResidualsPipeline(["a", "b"], scales, use_lcm=False, controlnets=nets, device=DEV)
    cn.prep_control_images({"a": [x for x in hints[0]], "b": [x for x in hints[1]]}, do_classifier_free_guidance=False, guess_mode=False)
    down, mid = cn(sample.to(DEV), 300, ehs.to(DEV), f, do_classifier_free_guidance=False, guess_mode=False)
    out = unet(sample.to(DEV), 300, ehs.to(DEV), down_block_additional_residuals=down, mid_block_additional_residual=mid).sample
    torch.cuda.synchronize()
    r = rel(out, ref)
    print("non-square 12-frame eps rel_l2 %.3e" % r)
    assert out.shape == ref.shape and r < 1e-2, r
