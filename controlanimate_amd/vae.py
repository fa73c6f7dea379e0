"""AutoencoderKL (SD1.5 VAE) on the HIP kernels: the encode/decode that brackets every window.

Mirrors the interface the reference uses from diffusers==0.23.0 `AutoencoderKL`
(animatediff/pipelines/controlanimation_pipeline.py:501-514 decode_latents, :566-588 encode in
prepare_latents; loaded at modules/controlanimate_pipeline.py:38-40):

    vae.encode(image[B,3,H,W] in [-1,1]).latent_dist.sample(generator) -> [B,4,H/8,W/8]
    vae.decode(z[B,4,h,w]).sample                                       -> [B,3,8h,8w]
    vae.config.scaling_factor, vae.dtype, vae.to(), vae.half()

State-dict keys are diffusers' (`encoder.down_blocks.0.resnets.0.conv1.weight`, ...,
`decoder.mid_block.attentions.0.to_q.weight`); the deprecated attention names written by the
reference's LDM converter (query/key/value/proj_attn, animatediff/utils/convert_from_ckpt.py:122-149,
627-629) are renamed on load like diffusers does.

Execution: channels-last fp16/bf16 activations, the UNet's kernels (ca_conv3x3 incl. the fused nearest
x2 upsample and the asymmetric-pad stride-2 downsample, ca_groupnorm_*+SiLU, ca_gemm); the mid
block's single-head head_dim-512 attention is scores-GEMM -> ca_softmax_rows -> PV-GEMM per image.
All frames of a window go through in one batch (the reference loops frame by frame; the VAE has no
cross-frame operation, so the results are the same).  There is no CPU fallback.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional, Tuple

import torch
from torch import nn

from . import kernels as K
from .layers import HipConv1x1, HipConv3x3, HipGroupNorm, HipLinear, WeightArena, _f32

VAE_CONFIG = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                  layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215, sample_size=512, act_fn="silu",
                  down_block_types=("DownEncoderBlock2D",) * 4, up_block_types=("UpDecoderBlock2D",) * 4)
EPS = 1e-6


class _ConvPadOut(HipConv3x3):
    """3x3 conv whose Cout is not a multiple of 4 (decoder.conv_out, 3 channels): packed with zero rows."""

    def pack(self, arena: WeightArena, dtype):
        npad = (self.out_channels + 3) // 4 * 4
        self.w = arena.add((npad, 3, 3, self.cin_pad), dtype,
                           lambda: torch.nn.functional.pad(self._packed_weight(), (0, 0, 0, 0, 0, 0, 0, npad - self.out_channels)))
        self.b = arena.add((npad,), torch.float32, lambda: torch.nn.functional.pad(_f32(self.bias), (0, npad - self.out_channels)))


class _Conv1x1Pad(HipConv1x1):
    """1x1 conv with both channel counts zero-padded to a multiple of 8 (post_quant_conv, 4 -> 4)."""

    def pack(self, arena: WeightArena, dtype):
        kp, np_ = (self.in_channels + 7) // 8 * 8, (self.out_channels + 7) // 8 * 8
        self.w = arena.add((np_, kp), dtype, lambda: torch.nn.functional.pad(
            _f32(self.weight).reshape(self.out_channels, self.in_channels), (0, kp - self.in_channels, 0, np_ - self.out_channels)))
        self.b = arena.add((np_,), torch.float32, lambda: torch.nn.functional.pad(_f32(self.bias), (0, np_ - self.out_channels)))


class ResnetBlock2D(nn.Module):
    """diffusers ResnetBlock2D(temb_channels=None, eps=1e-6, output_scale_factor=1)."""

    def __init__(self, cin: int, cout: int, groups: int):
        super().__init__()
        self.norm1 = HipGroupNorm(groups, cin, EPS)
        self.conv1 = HipConv3x3(cin, cout)
        self.norm2 = HipGroupNorm(groups, cout, EPS)
        self.conv2 = HipConv3x3(cout, cout)
        self.conv_shortcut = HipConv1x1(cin, cout) if cin != cout else None

    def run(self, x: torch.Tensor) -> torch.Tensor:
        h = self.conv1.run(self.norm1.run(x, act=K.ACT_SILU))
        h = self.norm2.run(h, act=K.ACT_SILU)
        if self.conv_shortcut is not None:
            n, hh, ww, c = x.shape
            x = self.conv_shortcut.run(x.view(n * hh * ww, c)).view(n, hh, ww, -1)
        return self.conv2.run(h, residual=x)


class VaeAttention(nn.Module):
    """diffusers Attention(query_dim=C, heads=1, dim_head=C, norm_num_groups=32, eps=1e-6, bias=True,
    residual_connection=True) as used by UNetMidBlock2D in the VAE."""

    def __init__(self, channels: int, groups: int):
        super().__init__()
        self.channels = channels
        self.group_norm = HipGroupNorm(groups, channels, EPS)
        self.to_q = HipLinear(channels, channels)
        self.to_k = HipLinear(channels, channels)
        self.to_v = HipLinear(channels, channels)
        self.to_out = nn.ModuleList([HipLinear(channels, channels), nn.Identity()])

    def run(self, x: torch.Tensor) -> torch.Tensor:
        n, hh, ww, c = x.shape
        hw = hh * ww
        xn = self.group_norm.run(x).view(n * hw, c)
        q, k = self.to_q.run(xn), self.to_k.run(xn)
        o = torch.empty_like(q)
        scale = c ** -0.5
        for i in range(n):
            rows = slice(i * hw, (i + 1) * hw)
            s = K.gemm(q[rows], k[rows], alpha=scale, out_f32=True)           # [hw, hw] fp32 scores
            p = K.softmax_rows(s, x.dtype)
            vt = K.gemm(self.to_v.w.t, xn[rows])                               # V^T [c, hw] (bias below)
            # rows of P sum to 1, so P (V + 1 b^T) = P V + b^T: to_v's bias is the column bias here
            K.gemm(p, vt, bias=self.to_v.b.t, out=o[rows])
        out = self.to_out[0].run(o, residual=x.view(n * hw, c))
        return out.view(n, hh, ww, c)


class MidBlock2D(nn.Module):
    def __init__(self, channels: int, groups: int):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(channels, channels, groups), ResnetBlock2D(channels, channels, groups)])
        self.attentions = nn.ModuleList([VaeAttention(channels, groups)])

    def run(self, x):
        return self.resnets[1].run(self.attentions[0].run(self.resnets[0].run(x)))


class _Sampler(nn.Module):
    def __init__(self, channels: int, stride: int):
        super().__init__()
        self.conv = HipConv3x3(channels, channels, stride=stride)


class DownEncoderBlock2D(nn.Module):
    def __init__(self, cin, cout, layers, groups, add_downsample):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if j == 0 else cout, cout, groups) for j in range(layers)])
        self.downsamplers = nn.ModuleList([_Sampler(cout, 2)]) if add_downsample else None

    def run(self, x):
        for r in self.resnets:
            x = r.run(x)
        if self.downsamplers is not None:  # Downsample2D(padding=0): F.pad(x, (0,1,0,1)) + conv stride 2
            x = self.downsamplers[0].conv.run(x, pad_asym=True)
        return x


class UpDecoderBlock2D(nn.Module):
    def __init__(self, cin, cout, layers, groups, add_upsample):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if j == 0 else cout, cout, groups) for j in range(layers)])
        self.upsamplers = nn.ModuleList([_Sampler(cout, 1)]) if add_upsample else None

    def run(self, x):
        for r in self.resnets:
            x = r.run(x)
        if self.upsamplers is not None:  # Upsample2D: nearest x2 fused into the conv's gather
            x = self.upsamplers[0].conv.run(x, upsample=True)
        return x


class Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        boc, g = cfg.block_out_channels, cfg.norm_num_groups
        self.conv_in = HipConv3x3(cfg.in_channels, boc[0])
        blocks, cin = [], boc[0]
        for i, cout in enumerate(boc):
            blocks.append(DownEncoderBlock2D(cin, cout, cfg.layers_per_block, g, i < len(boc) - 1))
            cin = cout
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = MidBlock2D(boc[-1], g)
        self.conv_norm_out = HipGroupNorm(g, boc[-1], EPS)
        self.conv_out = HipConv3x3(boc[-1], 2 * cfg.latent_channels)

    def run(self, x):
        x = self.conv_in.run(x)
        for b in self.down_blocks:
            x = b.run(x)
        x = self.mid_block.run(x)
        return self.conv_out.run(self.conv_norm_out.run(x, act=K.ACT_SILU))


class Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        rev, g = tuple(reversed(cfg.block_out_channels)), cfg.norm_num_groups
        self.conv_in = HipConv3x3(cfg.latent_channels, rev[0])
        self.mid_block = MidBlock2D(rev[0], g)
        blocks, cin = [], rev[0]
        for i, cout in enumerate(rev):
            blocks.append(UpDecoderBlock2D(cin, cout, cfg.layers_per_block + 1, g, i < len(rev) - 1))
            cin = cout
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = HipGroupNorm(g, rev[-1], EPS)
        self.conv_out = _ConvPadOut(rev[-1], cfg.out_channels)

    def run(self, x):
        x = self.mid_block.run(self.conv_in.run(x))
        for b in self.up_blocks:
            x = b.run(x)
        return self.conv_out.run(self.conv_norm_out.run(x, act=K.ACT_SILU), out_f32=True)


class DiagonalGaussianDistribution:
    """diffusers.models.vae.DiagonalGaussianDistribution over (mean, logvar) [B,4,h,w] fp32."""

    def __init__(self, mean: torch.Tensor, logvar: torch.Tensor):
        self.mean = mean
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        # randn_tensor: a CPU generator draws on the CPU and the sample is moved to the device
        gdev = generator.device if generator is not None else self.mean.device
        noise = torch.randn(self.mean.shape, generator=generator, device=gdev, dtype=torch.float32).to(self.mean.device)
        return self.mean + self.std * noise

    def mode(self) -> torch.Tensor:
        return self.mean


class AutoencoderKL(nn.Module):
    def __init__(self, **config):
        super().__init__()
        cfg = dict(VAE_CONFIG)
        cfg.update(config)
        cfg["block_out_channels"] = tuple(cfg["block_out_channels"])
        self.config = SimpleNamespace(**cfg)
        c = self.config
        self.encoder = Encoder(c)
        self.decoder = Decoder(c)
        self.quant_conv = HipConv1x1(2 * c.latent_channels, 2 * c.latent_channels)
        self.post_quant_conv = _Conv1x1Pad(c.latent_channels, c.latent_channels)
        self.arena: Optional[WeightArena] = None
        self.act_dtype = torch.float16  # the reference runs the VAE in fp16 (controlanimate_pipeline.py:108-110)
        self._register_load_state_dict_pre_hook(self._rename_deprecated_attention)

    @classmethod
    def from_config(cls, config: Optional[dict] = None, **kw):
        cfg = dict(config or {})
        cfg.update(kw)
        return cls(**{k: v for k, v in cfg.items() if not k.startswith("_")})

    @staticmethod
    def _rename_deprecated_attention(state_dict, prefix, *_):
        """query/key/value/proj_attn -> to_q/to_k/to_v/to_out.0 (diffusers _convert_deprecated_attention_blocks)."""
        ren = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}
        for k in list(state_dict.keys()):
            parts = k.split(".")
            if len(parts) >= 2 and "attentions" in parts and parts[-2] in ren:
                w = state_dict.pop(k)
                if w.dim() == 4:  # LDM checkpoints store the projections as 1x1 convs
                    w = w[:, :, 0, 0]
                state_dict[".".join(parts[:-2] + [ren[parts[-2]], parts[-1]])] = w

    # ---- device / dtype plumbing (the attributes the reference touches) ---------------------------
    @property
    def dtype(self):
        return self.act_dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def half(self):
        self.act_dtype, self.arena = torch.float16, None
        return self

    def bfloat16(self):
        self.act_dtype, self.arena = torch.bfloat16, None
        return self

    def to(self, *args, **kw):
        for a in list(args) + list(kw.values()):
            if a in (torch.float16, torch.bfloat16):
                self.act_dtype = a
        args = tuple(a for a in args if not isinstance(a, torch.dtype))
        kw = {k: v for k, v in kw.items() if not isinstance(v, torch.dtype)}
        if args or kw:
            super().to(*args, **kw)
            self.arena = None
        return self

    def prepare(self, device=None, dtype: Optional[torch.dtype] = None) -> "AutoencoderKL":
        if dtype is not None:
            self.act_dtype = dtype
        device = torch.device(device if device is not None else self.device)
        if device.type != "cuda":
            raise RuntimeError("prepare() needs a HIP device: the execution path has no CPU fallback")
        K.lib()
        arena = WeightArena()

        def walk(m):
            if hasattr(m, "pack"):
                m.pack(arena, self.act_dtype)
                return
            for ch in m.children():
                walk(ch)

        for ch in self.children():
            walk(ch)
        arena.finalize(device)
        self.arena = arena
        return self

    def _ready(self, device):
        if self.arena is None or self.arena.buffer.device != device:
            self.prepare(device)

    @staticmethod
    def _chunk(n: int, h: int, w: int) -> int:
        # the largest activation (256 channels at full resolution) must stay below the 4 GiB a buffer
        # descriptor of the LDS-DMA loads can address
        return max(1, min(n, int(3.5e9 // (h * w * 512))))

    # ---- API ---------------------------------------------------------------------------------------
    @torch.no_grad()
    def encode_moments(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """x [B,3,H,W] in [-1,1] -> (mean, logvar) fp32 [B,4,H/8,W/8] (logvar unclamped)."""
        if x.dim() != 4 or x.shape[1] != self.config.in_channels or x.shape[2] % 8 or x.shape[3] % 8:
            raise ValueError(f"expected [B,{self.config.in_channels},H,W] with H, W multiples of 8, got {tuple(x.shape)}")
        dev = x.device if x.is_cuda else self.device
        self._ready(dev)
        x = x.to(dev)
        lc = self.config.latent_channels
        outs = []
        step = self._chunk(x.shape[0], x.shape[2], x.shape[3])
        for s in range(0, x.shape[0], step):
            xb = x[s:s + step]
            nhwc = K.ncfhw_to_nhwc(xb.unsqueeze(0).permute(0, 2, 1, 3, 4), self.encoder.conv_in.cin_pad, self.act_dtype)
            h = self.encoder.run(nhwc)                                            # [n, H/8, W/8, 2*lc]
            n, hh, ww, c = h.shape
            m = self.quant_conv.run(h.view(n * hh * ww, c), out_f32=True).view(n, hh, ww, c)
            outs.append(m.permute(0, 3, 1, 2))
        mom = torch.cat(outs) if len(outs) > 1 else outs[0]
        return mom[:, :lc].contiguous(), mom[:, lc:].contiguous()

    def encode(self, x: torch.Tensor, return_dict: bool = True):
        mean, logvar = self.encode_moments(x)
        dist = DiagonalGaussianDistribution(mean, logvar)
        return SimpleNamespace(latent_dist=dist) if return_dict else (dist,)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True):
        """z [B,4,h,w] (already divided by scaling_factor) -> sample [B,3,8h,8w] fp32."""
        if z.dim() != 4 or z.shape[1] != self.config.latent_channels:
            raise ValueError(f"expected [B,{self.config.latent_channels},h,w], got {tuple(z.shape)}")
        dev = z.device if z.is_cuda else self.device
        self._ready(dev)
        z = z.to(dev)
        oc = self.config.out_channels
        outs = []
        step = self._chunk(z.shape[0], 8 * z.shape[2], 8 * z.shape[3])
        for s in range(0, z.shape[0], step):
            zb = z[s:s + step].float()
            x = K.ncfhw_to_nhwc(zb.unsqueeze(0).permute(0, 2, 1, 3, 4), 8, self.act_dtype)          # [n, h, w, 8]
            n, hh, ww, c = x.shape
            x = self.post_quant_conv.run(x.view(n * hh * ww, c)).view(n, hh, ww, -1)             # [n, h, w, 8]
            y = self.decoder.run(x)                                                               # [n, H, W, 4] fp32
            outs.append(y.permute(0, 3, 1, 2)[:, :oc])
        sample = torch.cat(outs) if len(outs) > 1 else outs[0]
        return SimpleNamespace(sample=sample) if return_dict else (sample,)
