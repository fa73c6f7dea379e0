"""ControlNet body pinned by the REFERENCE's own blocks (container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_controlnet_golden.py

diffusers' ControlNetModel is absent, but its encoder + mid block are the SD1.5 UNet encoder, and the reference
carries that arithmetic itself: animatediff/models/unet_blocks.py `CrossAttnDownBlock3D` / `DownBlock3D` /
`UNetMidBlock3DCrossAttn` (:283-523, :173-280) with `use_motion_module=False` and one frame are, per image, exactly
diffusers' 2-D blocks (ResnetBlock3D = inflated 2-D resnet, Transformer3DModel = Transformer2DModel over (b f)).
This script assembles a ControlNet from those reference blocks -- the only glue outside them is what SURVEY App. A-5
lists as plain convolutions: conv_in, the hint embedding (conv3x3 + SiLU chain), the 13 zero-convs (1x1), and the time
embedding of the import shim -- loads the oracle's synthetic weights under the diffusers key names, runs it on CPU and
stores inputs / the 13 residuals.  tests/test_oracle_golden.py checks oracle/controlnet.py against the file.
"""
from __future__ import annotations

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
import torch.nn.functional as F  # noqa: E402
from torch import nn  # noqa: E402

from oracle.controlnet import ControlNetConfig, init_controlnet_weights  # noqa: E402

from animatediff.models.resnet import InflatedConv3d  # noqa: E402  (reference)
from animatediff.models.unet_blocks import CrossAttnDownBlock3D, DownBlock3D, UNetMidBlock3DCrossAttn  # noqa: E402
from diffusers.models.embeddings import TimestepEmbedding, Timesteps  # noqa: E402  (the shim's restatement, as for the UNet fixtures)
from modules.attention_processor import CNAttnProcessor2_0  # noqa: E402


class RefControlNet(nn.Module):
    """SD1.5 ControlNet wired from the reference's 3-D blocks (f = 1)."""

    def __init__(self, cfg: ControlNetConfig):
        super().__init__()
        boc = cfg.block_out_channels
        temb = boc[0] * 4
        self.conv_in = InflatedConv3d(cfg.in_channels, boc[0], kernel_size=3, padding=1)
        self.time_proj = Timesteps(boc[0], True, 0)
        self.time_embedding = TimestepEmbedding(boc[0], temb)
        ce = cfg.cond_embedding_channels
        emb = nn.Module()
        emb.conv_in = InflatedConv3d(cfg.conditioning_channels, ce[0], kernel_size=3, padding=1)
        blocks = []
        for i in range(len(ce) - 1):
            blocks.append(InflatedConv3d(ce[i], ce[i], kernel_size=3, padding=1))
            blocks.append(InflatedConv3d(ce[i], ce[i + 1], kernel_size=3, padding=1, stride=2))
        emb.blocks = nn.ModuleList(blocks)
        emb.conv_out = InflatedConv3d(ce[-1], boc[0], kernel_size=3, padding=1)
        self.controlnet_cond_embedding = emb
        common = dict(temb_channels=temb, num_layers=cfg.layers_per_block, resnet_eps=cfg.norm_eps, resnet_act_fn="silu",
                      resnet_groups=cfg.norm_num_groups, use_inflated_groupnorm=False, use_motion_module=False,
                      motion_module_type=None, motion_module_kwargs=None)
        downs, zeros = [], [InflatedConv3d(boc[0], boc[0], kernel_size=1)]
        ch = boc[0]
        for i, co in enumerate(boc):
            last = i == len(boc) - 1
            if cfg.down_has_attn[i]:
                blk = CrossAttnDownBlock3D(in_channels=ch, out_channels=co, add_downsample=not last, downsample_padding=1,
                                           attn_num_head_channels=cfg.attention_heads, cross_attention_dim=cfg.cross_attention_dim,
                                           unet_use_cross_frame_attention=False, unet_use_temporal_attention=False, **common)
            else:
                blk = DownBlock3D(in_channels=ch, out_channels=co, add_downsample=not last, downsample_padding=1, **common)
            downs.append(blk)
            for _ in range(cfg.layers_per_block + (0 if last else 1)):
                zeros.append(InflatedConv3d(co, co, kernel_size=1))
            ch = co
        self.down_blocks = nn.ModuleList(downs)
        self.controlnet_down_blocks = nn.ModuleList(zeros)
        self.mid_block = UNetMidBlock3DCrossAttn(in_channels=boc[-1], temb_channels=temb, resnet_eps=cfg.norm_eps, resnet_act_fn="silu",
                                                 resnet_groups=cfg.norm_num_groups, attn_num_head_channels=cfg.attention_heads,
                                                 cross_attention_dim=cfg.cross_attention_dim, unet_use_cross_frame_attention=False,
                                                 unet_use_temporal_attention=False, use_inflated_groupnorm=False, use_motion_module=False,
                                                 motion_module_type=None, motion_module_kwargs=None)
        self.controlnet_mid_block = InflatedConv3d(boc[-1], boc[-1], kernel_size=1)

    def forward(self, sample, timestep, ehs, cond, scale, guess_mode):
        B = sample.shape[0]
        t = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1).expand(B)
        emb = self.time_embedding(self.time_proj(t).to(sample.dtype))
        x5, c5 = sample[:, :, None], cond[:, :, None]  # one frame per image
        e = F.silu(self.controlnet_cond_embedding.conv_in(c5))
        for blk in self.controlnet_cond_embedding.blocks:
            e = F.silu(blk(e))
        x = self.conv_in(x5) + self.controlnet_cond_embedding.conv_out(e)
        outs = [x]
        for blk in self.down_blocks:
            if getattr(blk, "has_cross_attention", False):
                x, res = blk(hidden_states=x, temb=emb, encoder_hidden_states=ehs)
            else:
                x, res = blk(hidden_states=x, temb=emb)
            outs += list(res)
        x = self.mid_block(x, emb, encoder_hidden_states=ehs)
        down = [z(o) for z, o in zip(self.controlnet_down_blocks, outs)]
        mid = self.controlnet_mid_block(x)
        if guess_mode:
            scales = torch.logspace(-1, 0, len(down) + 1) * scale
            down = [d * s for d, s in zip(down, scales)]
            mid = mid * scales[-1]
        else:
            down = [d * scale for d in down]
            mid = mid * scale
        return [d[:, :, 0] for d in down], mid[:, :, 0]


def wsum(sd) -> float:
    return float(sum(v.double().abs().sum().item() for v in sd.values()))


@torch.no_grad()
def main():
    torch.set_num_threads(8)
    out = {}
    for tag, boc, hw, B in (("w64", (64, 128, 256, 256), 8, 2), ("full", (320, 640, 1280, 1280), 4, 1)):
        cfg = ControlNetConfig(block_out_channels=boc)
        seed = 6 if tag == "w64" else 9
        w = init_controlnet_weights(cfg, seed=seed)
        m = RefControlNet(cfg).eval()
        # the reference blocks carry the unused block-level attention of BasicTransformerBlock
        missing, unexpected = m.load_state_dict(w, strict=False)
        assert not unexpected, unexpected[:4]
        assert all(".transformer_blocks.0.to_" in k for k in missing), missing[:4]
        g = torch.Generator().manual_seed(4242 + len(boc) + boc[0])
        x = torch.randn(B, 4, hw, hw, generator=g)
        ehs = torch.randn(B, 77, 768, generator=g) * 0.5
        cond = torch.rand(B, 3, 8 * hw, 8 * hw, generator=g)
        for mode, guess, scale, t in (("plain", False, 0.8, 750), ("guess", True, 1.0, 261)):
            down, mid = m(x, t, ehs, cond, scale, guess)
            for i, d in enumerate(down):
                out[f"{tag}_{mode}_down{i}"] = d
            out[f"{tag}_{mode}_mid"] = mid
            out[f"{tag}_{mode}_t"], out[f"{tag}_{mode}_scale"] = t, scale
        out.update({f"{tag}_sample": x, f"{tag}_ehs": ehs, f"{tag}_cond": cond, f"{tag}_seed": seed, f"{tag}_checksum": wsum(w)})
        if tag == "w64":
            # IP-Adapter run: the reference installs CNAttnProcessor2_0 on the ControlNets (modules/ip_adapter.py:129-134),
            # which drops the 4 image tokens of an 81-token context
            for mod in m.modules():
                if hasattr(mod, "set_processor") and hasattr(mod, "to_q"):
                    mod.set_processor(CNAttnProcessor2_0(num_tokens=4))
            ip_tokens = torch.randn(B, 4, 768, generator=g)
            ehs81 = torch.cat([ehs, ip_tokens], dim=1)
            down, mid = m(x, 500, ehs81, cond, 0.5, False)
            for i, d in enumerate(down):
                out[f"{tag}_cn_down{i}"] = d
            out[f"{tag}_cn_mid"], out[f"{tag}_cn_ip_tokens"] = mid, ip_tokens
    # the full-width tensors are large: keep the (small-spatial) residuals as fp16-exact? no -- fp32, but only mid + 3 downs
    keep = {}
    for k, v in out.items():
        if k.startswith("full_") and "_down" in k and not k.endswith(("down0", "down5", "down11")):
            continue
        keep[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    np.savez_compressed(os.path.join(HERE, "controlnet_refblocks.npz"), **keep)
    print("wrote controlnet_refblocks.npz", len(keep), "arrays", sum(v.nbytes for v in keep.values()) // 1024, "KB raw")


if __name__ == "__main__":
    if not os.path.isdir("/root/reference"):
        raise SystemExit("needs /root/reference (container only)")
    main()
