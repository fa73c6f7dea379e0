"""Localise a full-width mismatch: walk the first UNet ops on GPU and in the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from controlanimate_amd import kernels as K
from controlanimate_amd.configs import unet_config
from controlanimate_amd.context import ExecCtx
from controlanimate_amd.unet import UNet3DConditionModel
from oracle import nn_ops as ops
from oracle.unet3d import UNet3DConfig, init_unet3d_weights, resnet_block, transformer_block, motion_module, _conv5

DEV = "cuda:0"
ver = sys.argv[1] if len(sys.argv) > 1 else "v1"
cfg = UNet3DConfig.v1() if ver == "v1" else UNet3DConfig.v2()
w = init_unet3d_weights(cfg, seed=0)
g = torch.Generator().manual_seed(1)
f, hw = 8, 32
x = torch.randn(2, 4, f, hw, hw, generator=g)
ehs = torch.randn(2, 77, 768, generator=g) * 0.5
m = UNet3DConditionModel.from_config(unet_config(ver))
m.load_state_dict(w, strict=False)
m.to(DEV).prepare(DEV, torch.float16)
def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item()
def to5(y, b, c):
    return K.nhwc_to_ncfhw_f32(y.contiguous(), b, c, f).cpu()
with torch.no_grad():
    t = torch.full((2,), 500.0)
    emb = ops.time_embedding(w, "time_embedding", ops.timestep_sinusoid(t, 320), None)
    temb = m._time_embedding(500, 2, torch.device(DEV))
    ctx = ExecCtx(b=2, f=f, dtype=torch.float16, temb=temb, emb_groups=2, ehs=ehs.to(DEV).half(), frames_per_kv=f,
                  gn_frames_per_stat=1 if cfg.use_inflated_groupnorm else f, cache={})
    r0 = m.down_blocks[0].resnets[0]
    lo, hi = r0.temb_slice
    ref_t = ops.linear(w, "down_blocks.0.resnets.0.time_emb_proj", F.silu(emb))
    print("temb proj rel", rel(temb[:, lo:hi], ref_t))
    xo = _conv5(w, "conv_in", x)
    xg = m.conv_in.run(K.ncfhw_to_nhwc(x.to(DEV), 8, torch.float16))
    print("conv_in rel", rel(to5(xg, 2, 320), xo))
    for name, fn_o, mod in [
        ("down0.resnet0", lambda z: resnet_block(w, "down_blocks.0.resnets.0", z, emb, cfg), lambda z: m.down_blocks[0].resnets[0](z, ctx)),
        ("down0.attn0", lambda z: transformer_block(w, "down_blocks.0.attentions.0", z, ehs, cfg), lambda z: m.down_blocks[0].attentions[0](z, ctx)),
        ("down0.motion0", lambda z: motion_module(w, "down_blocks.0.motion_modules.0", z, cfg), lambda z: m.down_blocks[0].motion_modules[0](z, ctx)),
        ("down0.resnet1", lambda z: resnet_block(w, "down_blocks.0.resnets.1", z, emb, cfg), lambda z: m.down_blocks[0].resnets[1](z, ctx)),
        ("down0.attn1", lambda z: transformer_block(w, "down_blocks.0.attentions.1", z, ehs, cfg), lambda z: m.down_blocks[0].attentions[1](z, ctx)),
        ("down0.motion1", lambda z: motion_module(w, "down_blocks.0.motion_modules.1", z, cfg), lambda z: m.down_blocks[0].motion_modules[1](z, ctx)),
        ("down0.downsample", lambda z: _conv5(w, "down_blocks.0.downsamplers.0.conv", z, stride=2), lambda z: m.down_blocks[0].downsamplers[0](z)),
        ("down1.resnet0", lambda z: resnet_block(w, "down_blocks.1.resnets.0", z, emb, cfg), lambda z: m.down_blocks[1].resnets[0](z, ctx)),
        ("down1.attn0", lambda z: transformer_block(w, "down_blocks.1.attentions.0", z, ehs, cfg), lambda z: m.down_blocks[1].attentions[0](z, ctx)),
        ("down1.motion0", lambda z: motion_module(w, "down_blocks.1.motion_modules.0", z, cfg), lambda z: m.down_blocks[1].motion_modules[0](z, ctx)),
    ]:
        xo_new = fn_o(xo)
        # feed the GPU module with the ORACLE's input (rounded) to isolate per-module error, and also chain
        xin = K.ncfhw_to_nhwc(xo.to(DEV), xo.shape[1], torch.float16)
        yg_iso = mod(xin)
        xg = mod(xg)
        c = xo_new.shape[1]
        print(f"{name:18s} isolated rel {rel(to5(yg_iso, 2, c), xo_new):.3e}   chained rel {rel(to5(xg, 2, c), xo_new):.3e}   |ref| max {xo_new.abs().max():.2f}")
        xo = xo_new
