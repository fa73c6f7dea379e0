"""Generates tests/golden/*.npz|json by running the REFERENCE's own modules (container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Needs /root/reference (read-only) and the import shim in _refstub.py; it refuses to run elsewhere.
The fixtures hold inputs and expected outputs only (data); weights are seeded synthetic tensors
re-created on the test side by oracle.unet3d.init_from_shapes (a checksum of them is stored so RNG
drift is detected instead of mis-reported as a parity failure).
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _refstub  # noqa: E402

_refstub.install()

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

from oracle.unet3d import (UNet3DConfig, init_from_shapes, init_unet3d_weights, sub_state_dict,  # noqa: E402
                           unet3d_param_shapes)

from animatediff.models.unet import UNet3DConditionModel  # noqa: E402  (reference)
from animatediff.models.resnet import ResnetBlock3D  # noqa: E402
from animatediff.models.attention import Transformer3DModel  # noqa: E402
from animatediff.models.motion_module import VanillaTemporalModule  # noqa: E402
from modules.attention_processor import AttnProcessor2_0, CNAttnProcessor2_0, IPAttnProcessor2_0  # noqa: E402

SMALL = (64, 128, 256, 256)


def wsum(sd) -> float:
    """Order-independent checksum of a weight dict."""
    return float(sum(v.double().abs().sum().item() for v in sd.values()))


def ref_unet(cfg: UNet3DConfig, version: str):
    y = yaml.safe_load(open(f"/root/reference/configs/inference/inference-{version}.yaml"))["unet_additional_kwargs"]
    sd15 = dict(sample_size=64, in_channels=4, out_channels=4,
                down_block_types=("CrossAttnDownBlock3D",) * 3 + ("DownBlock3D",),
                up_block_types=("UpBlock3D",) + ("CrossAttnUpBlock3D",) * 3,
                block_out_channels=cfg.block_out_channels, layers_per_block=2, cross_attention_dim=768,
                attention_head_dim=8, time_cond_proj_dim=cfg.time_cond_proj_dim)
    return UNet3DConditionModel.from_config(sd15, **y).eval()


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, {k: tuple(np.shape(v)) for k, v in out.items()})


@torch.no_grad()
def unet_fixtures():
    g = torch.Generator().manual_seed(1234)
    # ---- v2 (per-frame GN, mid motion module, PE 32), 16 frames, CFG batch 2
    cfg = UNet3DConfig.v2(block_out_channels=SMALL)
    w = init_unet3d_weights(cfg, seed=1)
    m = ref_unet(cfg, "v2")
    missing, unexpected = m.load_state_dict(w, strict=False)
    assert not unexpected and all(".transformer_blocks.0.to_" in k for k in missing), (missing[:3], unexpected[:3])
    x = torch.randn(2, 4, 16, 8, 8, generator=g)
    ehs = torch.randn(2, 77, 768, generator=g) * 0.5
    out = m(x, 500, ehs).sample
    save("unet3d_v2_w64.npz", sample=x, ehs=ehs, timestep=500, out=out, weight_seed=1, weight_checksum=wsum(w))
    procs_v2 = list(m.attn_processors.keys())

    # ---- IP-Adapter processors installed on the same model (rule of modules/ip_adapter.py:95-127)
    gi = torch.Generator().manual_seed(77)
    ipw = {}
    procs = {}
    for name in m.attn_processors.keys():
        is_plain = name.endswith("attn1.processor") or "temporal_transformer" in name or "attn" not in name
        if is_plain:
            procs[name] = AttnProcessor2_0()
            continue
        if name.startswith("mid_block"):
            hidden = cfg.block_out_channels[-1]
        elif name.startswith("up_blocks"):
            hidden = list(reversed(cfg.block_out_channels))[int(name[len("up_blocks.")])]
        else:
            hidden = cfg.block_out_channels[int(name[len("down_blocks.")])]
        p = IPAttnProcessor2_0(hidden_size=hidden, cross_attention_dim=768, scale=0.6, num_tokens=4)
        p.to_k_ip.weight.copy_(torch.randn(hidden, 768, generator=gi) * 768 ** -0.5)
        p.to_v_ip.weight.copy_(torch.randn(hidden, 768, generator=gi) * 768 ** -0.5)
        ipw[name] = p
        procs[name] = p
    m.set_attn_processor(procs)
    ehs_ip = torch.cat([ehs, torch.randn(2, 4, 768, generator=g) * 0.5], dim=1)
    out_ip = m(x, 261, ehs_ip).sample
    save("unet3d_ip_w64.npz", sample=x, ehs=ehs_ip, timestep=261, out=out_ip, weight_seed=1, weight_checksum=wsum(w),
         ip_seed=77, ip_scale=0.6, ip_sites=np.array(list(ipw.keys())),
         ip_checksum=float(sum(p.to_k_ip.weight.double().abs().sum().item() + p.to_v_ip.weight.double().abs().sum().item() for p in ipw.values())))

    # ---- v1 (cross-frame GN, no mid motion module, PE 24), 8 frames, ControlNet-style residuals with b=1
    cfg1 = UNet3DConfig.v1(block_out_channels=SMALL)
    w1 = init_unet3d_weights(cfg1, seed=2)
    m1 = ref_unet(cfg1, "v1")
    m1.load_state_dict(w1, strict=False)
    x1 = torch.randn(2, 4, 8, 8, 8, generator=g)
    ehs1 = torch.randn(2, 77, 768, generator=g) * 0.5
    chans = [64, 64, 64, 64, 128, 128, 128, 256, 256, 256, 256, 256]
    sizes = [8, 8, 8, 4, 4, 4, 2, 2, 2, 1, 1, 1]
    down = [torch.randn(1, c, 8, s, s, generator=g) * 0.3 for c, s in zip(chans, sizes)]
    mid = torch.randn(1, 256, 8, 1, 1, generator=g) * 0.3
    out1 = m1(x1, torch.tensor(981), ehs1, down_block_additional_residuals=tuple(down), mid_block_additional_residual=mid).sample
    arrs = {f"down{i}": d for i, d in enumerate(down)}
    save("unet3d_v1_w64.npz", sample=x1, ehs=ehs1, timestep=981, out=out1, mid=mid, weight_seed=2, weight_checksum=wsum(w1), **arrs)
    procs_v1 = list(m1.attn_processors.keys())

    # ---- native-LCM UNet (time_cond_proj_dim=256) with the reference's w-embedding
    from animatediff.pipelines.controlanimation_pipeline import ControlAnimationPipeline  # reference
    cfgl = UNet3DConfig.v2(block_out_channels=SMALL, time_cond_proj_dim=256)
    wl = init_unet3d_weights(cfgl, seed=3)
    ml = ref_unet(cfgl, "v2")
    ml.load_state_dict(wl, strict=False)
    wemb = ControlAnimationPipeline.get_w_embedding(None, torch.tensor([7.5]), embedding_dim=256)
    xl = torch.randn(1, 4, 16, 8, 8, generator=g)
    ehsl = torch.randn(1, 77, 768, generator=g) * 0.5
    outl = ml(xl, torch.full((1,), 499, dtype=torch.long), ehsl, timestep_cond=wemb).sample
    save("unet3d_lcm_w64.npz", sample=xl, ehs=ehsl, timestep=499, w_embedding=wemb, out=outl, weight_seed=3, weight_checksum=wsum(wl))
    return procs_v1, procs_v2


@torch.no_grad()
def module_fixtures():
    """Single modules at REAL SD1.5 widths (head dims 40 / 80 / 160) on tiny spatial sizes."""
    g = torch.Generator().manual_seed(4321)
    full = UNet3DConfig.v2()
    shapes = unet3d_param_shapes(full)
    out = {}
    # ResnetBlock3D 320 -> 640 with shortcut, per-frame and cross-frame GroupNorm
    pre = "down_blocks.1.resnets.0"
    sh = {k[len(pre) + 1:]: v for k, v in shapes.items() if k.startswith(pre + ".")}
    w = init_from_shapes(sh, seed=11)
    x = torch.randn(2, 320, 4, 6, 6, generator=g)
    temb = torch.randn(2, 1280, generator=g)
    for infl in (True, False):
        r = ResnetBlock3D(in_channels=320, out_channels=640, temb_channels=1280, eps=1e-5, groups=32,
                          non_linearity="silu", use_inflated_groupnorm=infl).eval()
        r.load_state_dict(w)
        out[f"resnet_out_inflated{int(infl)}"] = r(x, temb)
    out.update(resnet_x=x, resnet_temb=temb, resnet_seed=11, resnet_checksum=wsum(w))
    # Transformer3DModel C=320 (d=40), C=640 (d=80), C=1280 (d=160)
    for name, c, pre in (("tx320", 320, "down_blocks.0.attentions.0"), ("tx640", 640, "down_blocks.1.attentions.0"),
                         ("tx1280", 1280, "down_blocks.2.attentions.0")):
        sh = {k[len(pre) + 1:]: v for k, v in shapes.items() if k.startswith(pre + ".")}
        seed = 20 + c // 320
        w = init_from_shapes(sh, seed=seed)
        t = Transformer3DModel(8, c // 8, in_channels=c, num_layers=1, cross_attention_dim=768, norm_num_groups=32,
                               unet_use_cross_frame_attention=False, unet_use_temporal_attention=False).eval()
        missing, unexpected = t.load_state_dict(w, strict=False)
        assert not unexpected
        x = torch.randn(1, c, 3, 6, 4, generator=g)
        ehs = torch.randn(1, 77, 768, generator=g) * 0.5
        out.update({f"{name}_x": x, f"{name}_ehs": ehs, f"{name}_out": t(x, encoder_hidden_states=ehs).sample,
                    f"{name}_seed": seed, f"{name}_checksum": wsum(w)})
    # VanillaTemporalModule C=320 / 640 / 1280, 16 frames (PE 32) and 8 frames
    mm_kwargs = dict(num_attention_heads=8, num_transformer_block=1, attention_block_types=["Temporal_Self", "Temporal_Self"],
                     temporal_position_encoding=True, temporal_position_encoding_max_len=32, temporal_attention_dim_div=1)
    for name, c, pre, f in (("mm320", 320, "down_blocks.0.motion_modules.0", 16), ("mm640", 640, "down_blocks.1.motion_modules.0", 8),
                            ("mm1280", 1280, "down_blocks.2.motion_modules.0", 16)):
        sh = {k[len(pre) + 1:]: v for k, v in shapes.items() if k.startswith(pre + ".")}
        seed = 30 + c // 320
        w = init_from_shapes(sh, seed=seed)
        mmod = VanillaTemporalModule(in_channels=c, **mm_kwargs).eval()
        mmod.load_state_dict(w)
        x = torch.randn(2, c, f, 3, 2, generator=g)
        out.update({f"{name}_x": x, f"{name}_out": mmod(x, None, None), f"{name}_seed": seed, f"{name}_checksum": wsum(w)})
    save("modules_fullwidth.npz", **out)


@torch.no_grad()
def processor_fixtures():
    """The reference's IP / CN processors on a bare Attention (modules/attention_processor.py)."""
    from diffusers.models.attention import Attention  # the shim class
    g = torch.Generator().manual_seed(99)
    c, heads = 320, 8
    attn = Attention(query_dim=c, cross_attention_dim=768, heads=heads, dim_head=c // heads).eval()
    sh = {"to_q.weight": (c, c), "to_k.weight": (c, 768), "to_v.weight": (c, 768), "to_out.0.weight": (c, c), "to_out.0.bias": (c,)}
    w = init_from_shapes(sh, seed=41)
    attn.load_state_dict(w)
    x = torch.randn(3, 20, c, generator=g)
    ctx = torch.randn(3, 81, 768, generator=g) * 0.5
    ip = IPAttnProcessor2_0(hidden_size=c, cross_attention_dim=768, scale=0.4, num_tokens=4)
    kip = torch.randn(c, 768, generator=g) * 768 ** -0.5
    vip = torch.randn(c, 768, generator=g) * 768 ** -0.5
    ip.to_k_ip.weight.copy_(kip)
    ip.to_v_ip.weight.copy_(vip)
    out_ip = ip(attn, x, encoder_hidden_states=ctx)
    out_cn = CNAttnProcessor2_0(num_tokens=4)(attn, x, encoder_hidden_states=ctx)
    out_plain = AttnProcessor2_0()(attn, x, encoder_hidden_states=ctx)
    save("attn_processors.npz", x=x, ctx=ctx, to_k_ip=kip, to_v_ip=vip, ip_scale=0.4, out_ip=out_ip, out_cn=out_cn,
         out_plain=out_plain, seed=41, checksum=wsum(w))


def scheduler_fixtures():
    """Known answers of the reference's in-tree LCMScheduler and get_w_embedding."""
    from animatediff.pipelines.controlanimation_pipeline import ControlAnimationPipeline, LCMScheduler
    s = LCMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", prediction_type="epsilon")
    arrs = {}
    combos = [(1.0, 20), (0.5, 4), (0.92, 20), (1.0, 4), (0.75, 8), (1.0, 1)]
    for k, (strength, steps) in enumerate(combos):
        s.set_timesteps(strength, steps, 50)
        arrs[f"timesteps_{k}"] = s.timesteps.numpy()
    arrs["combos"] = np.array(combos, dtype=np.float64)
    arrs["alphas_cumprod"] = s.alphas_cumprod.numpy()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 4, 8, 6, 6, generator=g)
    eps = torch.randn(1, 4, 8, 6, 6, generator=g)
    s.set_timesteps(0.5, 4, 50)
    for i, t in enumerate(s.timesteps):
        torch.manual_seed(100 + i)  # the reference draws torch.randn from the GLOBAL CPU RNG (:1601)
        prev, den = s.step(eps, i, t, x, return_dict=False)
        arrs[f"step{i}_prev"], arrs[f"step{i}_denoised"] = prev.numpy(), den.numpy()
    arrs["step_sample"], arrs["step_eps"] = x.numpy(), eps.numpy()
    noise = torch.randn(1, 4, 8, 6, 6, generator=g)
    arrs["add_noise_t499"] = s.add_noise(x, noise, torch.tensor([499])).numpy()
    arrs["add_noise_noise"] = noise.numpy()
    for t in (999, 499, 19):
        cs, co = s.get_scalings_for_boundary_condition_discrete(t)
        arrs[f"scalings_{t}"] = np.array([cs, co], dtype=np.float64)
    arrs["w_embedding_7p5"] = ControlAnimationPipeline.get_w_embedding(None, torch.tensor([7.5]), embedding_dim=256).numpy()
    arrs["w_embedding_1p35"] = ControlAnimationPipeline.get_w_embedding(None, torch.tensor([1.35]), embedding_dim=256).numpy()
    np.savez_compressed(os.path.join(HERE, "lcm_custom.npz"), **arrs)
    print("wrote lcm_custom.npz", len(arrs))


def key_fixtures(procs_v1, procs_v2):
    """Checkpoint-key contract at FULL width (meta device: no memory), plus attn_processors order."""
    res = {}
    for version, cfg in (("v1", UNet3DConfig.v1()), ("v2", UNet3DConfig.v2())):
        with torch.device("meta"):
            m = ref_unet(cfg, version)
        sd = m.state_dict()
        res[version] = {"keys": {k: list(v.shape) for k, v in sd.items()},
                        "num_params": int(sum(p.numel() for p in m.parameters())),
                        "attn_processors": list(m.attn_processors.keys())}
        assert res[version]["attn_processors"] == (procs_v1 if version == "v1" else procs_v2)
    with open(os.path.join(HERE, "unet3d_keys.json"), "w") as f:
        json.dump(res, f, separators=(",", ":"))
    print("wrote unet3d_keys.json", {v: (len(res[v]["keys"]), res[v]["num_params"], len(res[v]["attn_processors"])) for v in res})


if __name__ == "__main__":
    torch.set_num_threads(8)
    p1, p2 = unet_fixtures()
    module_fixtures()
    processor_fixtures()
    scheduler_fixtures()
    key_fixtures(p1, p2)
