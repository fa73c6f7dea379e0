"""CPU: the oracle (fp32 restatement) against fixtures produced by the REFERENCE's own modules
(tests/golden/make_golden.py, run in the build container). This is what pins the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import schedulers as S
from oracle.nn_ops import attention as oracle_attention
from oracle.unet3d import (UNet3DConfig, init_from_shapes, init_unet3d_weights, motion_module, resnet_block,
                           transformer_block, unet3d_forward, unet3d_param_shapes)

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SMALL = (64, 128, 256, 256)


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def wsum(sd):
    return float(sum(v.double().abs().sum().item() for v in sd.values()))


def T(a):
    return torch.from_numpy(np.asarray(a))


def check_weights(sd, fx, key="weight_checksum"):
    assert abs(wsum(sd) - float(fx[key])) <= 1e-6 * float(fx[key]), "seeded weights differ from the fixture's (torch RNG drift)"


def assert_close(a, b, tol=2e-5):
    rel = ((a - b).norm() / b.norm()).item()
    assert rel < tol, f"rel_l2 {rel:.3e}"


def test_unet3d_v2_matches_reference():
    fx = load("unet3d_v2_w64.npz")
    cfg = UNet3DConfig.v2(block_out_channels=SMALL)
    w = init_unet3d_weights(cfg, seed=int(fx["weight_seed"]))
    check_weights(w, fx)
    out = unet3d_forward(w, cfg, T(fx["sample"]), int(fx["timestep"]), T(fx["ehs"]))
    assert_close(out, T(fx["out"]))


def test_unet3d_v1_cross_frame_groupnorm_and_controlnet_residual_broadcast():
    fx = load("unet3d_v1_w64.npz")
    cfg = UNet3DConfig.v1(block_out_channels=SMALL)
    w = init_unet3d_weights(cfg, seed=int(fx["weight_seed"]))
    check_weights(w, fx)
    down = [T(fx[f"down{i}"]) for i in range(12)]
    out = unet3d_forward(w, cfg, T(fx["sample"]), int(fx["timestep"]), T(fx["ehs"]), down, T(fx["mid"]))
    assert_close(out, T(fx["out"]))


def test_unet3d_native_lcm_timestep_cond():
    fx = load("unet3d_lcm_w64.npz")
    cfg = UNet3DConfig.v2(block_out_channels=SMALL, time_cond_proj_dim=256)
    w = init_unet3d_weights(cfg, seed=int(fx["weight_seed"]))
    check_weights(w, fx)
    wemb = S.get_w_embedding(torch.tensor([7.5]), 256)
    assert torch.allclose(wemb, T(fx["w_embedding"]), atol=2e-3)  # fp32 sin/cos at ~7500 rad: CPU-library dependent
    out = unet3d_forward(w, cfg, T(fx["sample"]), int(fx["timestep"]), T(fx["ehs"]), timestep_cond=T(fx["w_embedding"]))
    assert_close(out, T(fx["out"]))


def ip_weights_from_fixture(fx, cfg):
    """Re-creates the IP processors' weights in the order make_golden.py drew them."""
    g = torch.Generator().manual_seed(int(fx["ip_seed"]))
    ip = {}
    for name in [str(s) for s in fx["ip_sites"]]:
        if name.startswith("mid_block"):
            hidden = cfg.block_out_channels[-1]
        elif name.startswith("up_blocks"):
            hidden = list(reversed(cfg.block_out_channels))[int(name[len("up_blocks.")])]
        else:
            hidden = cfg.block_out_channels[int(name[len("down_blocks.")])]
        k = torch.randn(hidden, 768, generator=g) * 768 ** -0.5
        v = torch.randn(hidden, 768, generator=g) * 768 ** -0.5
        ip[name[: -len(".processor")]] = {"to_k_ip": k, "to_v_ip": v, "scale": float(fx["ip_scale"]), "num_tokens": 4}
    return ip


def test_unet3d_ip_adapter_sites():
    fx = load("unet3d_ip_w64.npz")
    cfg = UNet3DConfig.v2(block_out_channels=SMALL)
    w = init_unet3d_weights(cfg, seed=int(fx["weight_seed"]))
    ip = ip_weights_from_fixture(fx, cfg)
    assert len(ip) == 16
    cs = sum(d["to_k_ip"].double().abs().sum().item() + d["to_v_ip"].double().abs().sum().item() for d in ip.values())
    assert abs(cs - float(fx["ip_checksum"])) < 1e-6 * cs
    out = unet3d_forward(w, cfg, T(fx["sample"]), int(fx["timestep"]), T(fx["ehs"]), ip=ip)
    assert_close(out, T(fx["out"]))


def test_fullwidth_modules_head_dims_40_80_160():
    fx = load("modules_fullwidth.npz")
    full = UNet3DConfig.v2()
    shapes = unet3d_param_shapes(full)

    def sub(pre):
        return {k: v for k, v in shapes.items() if k.startswith(pre + ".")}

    pre = "down_blocks.1.resnets.0"
    w = {pre + "." + k: v for k, v in init_from_shapes({k[len(pre) + 1:]: v for k, v in sub(pre).items()}, seed=int(fx["resnet_seed"])).items()}
    assert abs(wsum(w) - float(fx["resnet_checksum"])) < 1e-6 * wsum(w)
    for infl in (1, 0):
        cfg = UNet3DConfig(use_inflated_groupnorm=bool(infl))
        out = resnet_block(w, pre, T(fx["resnet_x"]), T(fx["resnet_temb"]), cfg)
        assert_close(out, T(fx[f"resnet_out_inflated{infl}"]))
    for name, pre in (("tx320", "down_blocks.0.attentions.0"), ("tx640", "down_blocks.1.attentions.0"), ("tx1280", "down_blocks.2.attentions.0")):
        w = {pre + "." + k: v for k, v in init_from_shapes({k[len(pre) + 1:]: v for k, v in sub(pre).items()}, seed=int(fx[f"{name}_seed"])).items()}
        assert abs(wsum(w) - float(fx[f"{name}_checksum"])) < 1e-6 * wsum(w)
        out = transformer_block(w, pre, T(fx[f"{name}_x"]), T(fx[f"{name}_ehs"]), full)
        assert_close(out, T(fx[f"{name}_out"]))
    for name, pre in (("mm320", "down_blocks.0.motion_modules.0"), ("mm640", "down_blocks.1.motion_modules.0"), ("mm1280", "down_blocks.2.motion_modules.0")):
        w = {pre + "." + k: v for k, v in init_from_shapes({k[len(pre) + 1:]: v for k, v in sub(pre).items()}, seed=int(fx[f"{name}_seed"])).items()}
        assert abs(wsum(w) - float(fx[f"{name}_checksum"])) < 1e-6 * wsum(w)
        out = motion_module(w, pre, T(fx[f"{name}_x"]), full)
        assert_close(out, T(fx[f"{name}_out"]))


def test_attention_processors_ip_and_cn():
    fx = load("attn_processors.npz")
    c = 320
    sh = {"to_q.weight": (c, c), "to_k.weight": (c, 768), "to_v.weight": (c, 768), "to_out.0.weight": (c, c), "to_out.0.bias": (c,)}
    w = {"a." + k: v for k, v in init_from_shapes(sh, seed=int(fx["seed"])).items()}
    assert abs(wsum(w) - float(fx["checksum"])) < 1e-6 * wsum(w)
    x, ctx = T(fx["x"]), T(fx["ctx"])
    ip = {"to_k_ip": T(fx["to_k_ip"]), "to_v_ip": T(fx["to_v_ip"]), "scale": float(fx["ip_scale"]), "num_tokens": 4}
    assert_close(oracle_attention(w, "a", x, ctx, 8, ip=ip), T(fx["out_ip"]))
    assert_close(oracle_attention(w, "a", x, ctx, 8, strip_tokens=4), T(fx["out_cn"]))
    assert_close(oracle_attention(w, "a", x, ctx, 8), T(fx["out_plain"]))


def test_custom_lcm_scheduler_known_answers():
    fx = load("lcm_custom.npz")
    s = S.CustomLCM()
    assert np.allclose(s.alphas_cumprod.numpy(), fx["alphas_cumprod"], rtol=0, atol=0)
    for k, (strength, steps) in enumerate(fx["combos"]):
        s.set_timesteps(float(strength), int(steps), 50)
        assert np.array_equal(s.timesteps.numpy(), fx[f"timesteps_{k}"])
    # SURVEY App. D known answers
    s.set_timesteps(1.0, 20, 50)
    assert s.timesteps.tolist() == list(range(999, 238, -40))
    s.set_timesteps(0.5, 4, 50)
    assert s.timesteps.tolist() == [499, 379, 259, 139]
    x, eps = T(fx["step_sample"]), T(fx["step_eps"])
    for i, t in enumerate(s.timesteps):
        torch.manual_seed(100 + i)
        prev, den = s.step(eps, i, t, x)
        assert torch.allclose(prev, T(fx[f"step{i}_prev"]), atol=1e-6, rtol=1e-6)
        assert torch.allclose(den, T(fx[f"step{i}_denoised"]), atol=1e-6, rtol=1e-6)
    assert torch.allclose(s.add_noise(x, T(fx["add_noise_noise"]), torch.tensor([499])), T(fx["add_noise_t499"]), atol=1e-6)
    for t in (999, 499, 19):
        cs, co = S.CustomLCM.scalings(t)
        assert np.allclose([cs, co], fx[f"scalings_{t}"], rtol=1e-12)
    assert torch.allclose(S.get_w_embedding(torch.tensor([7.5]), 256), T(fx["w_embedding_7p5"]), atol=2e-3)
    assert torch.allclose(S.get_w_embedding(torch.tensor([1.35]), 256), T(fx["w_embedding_1p35"]), atol=5e-4)


def test_checkpoint_key_contract_full_width():
    with open(os.path.join(G, "unet3d_keys.json")) as f:
        ref = json.load(f)
    for version, cfg in (("v1", UNet3DConfig.v1()), ("v2", UNet3DConfig.v2())):
        mine = {k: list(v) for k, v in unet3d_param_shapes(cfg, include_dead=True).items()}
        assert mine == ref[version]["keys"], version


# ------------------------------------------------------------------------------------------------------------------
# ControlNet body: pinned by the reference's OWN blocks (tests/golden/make_controlnet_golden.py assembles an SD1.5
# ControlNet from animatediff/models/unet_blocks.py with use_motion_module=False, one frame per image).
@pytest.mark.parametrize("tag,boc", [("w64", SMALL), ("full", (320, 640, 1280, 1280))])
def test_controlnet_oracle_matches_reference_blocks(tag, boc):
    from oracle.controlnet import ControlNetConfig, controlnet_forward, init_controlnet_weights
    fx = load("controlnet_refblocks.npz")
    cfg = ControlNetConfig(block_out_channels=boc)
    w = init_controlnet_weights(cfg, seed=int(fx[f"{tag}_seed"]))
    check_weights(w, fx, f"{tag}_checksum")
    x, ehs, cond = T(fx[f"{tag}_sample"]), T(fx[f"{tag}_ehs"]), T(fx[f"{tag}_cond"])
    kept = range(12) if tag == "w64" else (0, 5, 11)
    for mode, guess in (("plain", False), ("guess", True)):
        down, mid = controlnet_forward(w, cfg, x, int(fx[f"{tag}_{mode}_t"]), ehs, cond, float(fx[f"{tag}_{mode}_scale"]), guess)
        assert len(down) == 12
        for i in kept:
            assert_close(down[i], T(fx[f"{tag}_{mode}_down{i}"]), 3e-5)
        assert_close(mid, T(fx[f"{tag}_{mode}_mid"]), 3e-5)
    if tag == "w64":  # CNAttnProcessor2_0 installed by the IP-Adapter path: the 4 image tokens are dropped
        ehs81 = torch.cat([ehs, T(fx["w64_cn_ip_tokens"])], dim=1)
        down, mid = controlnet_forward(w, cfg, x, 500, ehs81, cond, 0.5, False, strip_tokens=4)
        for i in range(12):
            assert_close(down[i], T(fx[f"w64_cn_down{i}"]), 3e-5)
        assert_close(mid, T(fx["w64_cn_mid"]), 3e-5)
