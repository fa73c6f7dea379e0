"""ControlNetModel / MultiControlNetModel (SD1.5 configuration) on the HIP kernels.

The reference owns no ControlNet arithmetic: it instantiates diffusers==0.23.0's classes
(modules/controlresiduals_pipeline.py:18-19,32-38) and calls MultiControlNetModel.forward (:294-302).
This file keeps that module API -- class names, `nets`, checkpoint keys (conv_in, time_embedding,
controlnet_cond_embedding.{conv_in,blocks.0-5,conv_out}, down_blocks.*, mid_block.*,
controlnet_down_blocks.0-11, controlnet_mid_block; cf. the reference's own converter
animatediff/utils/convert_from_ckpt.py:514-554), forward arguments, `set_attn_processor` -- over the
same NHWC kernels as the UNet.  Fusions:
  * the hint embedding (8 convs at up to 512x512) does not depend on the timestep: it is computed
    once per window and reused by every denoising step (the reference recomputes it each step);
  * zero-conv epilogue applies conditioning_scale (and the guess-mode logspace factor) and
    accumulates across nets: MultiControlNet's sum never exists as a separate pass;
  * conv_in's epilogue adds the hint embedding.
"""
from __future__ import annotations

import dataclasses
from collections import OrderedDict

from typing import List, Optional, Sequence, Tuple, Union

import torch
from torch import nn

from . import kernels as K
from .context import ExecCtx
from .layers import HipConv1x1, HipConv3x3
from .unet import FrozenConfig, HipModelMixin, TimestepEmbedding, Timesteps
from .unet_blocks import UNetMidBlock3DCrossAttn, get_down_block


class ControlNetConditioningEmbedding(nn.Module):
    def __init__(self, conditioning_embedding_channels: int, conditioning_channels: int = 3,
                 block_out_channels: Tuple[int, ...] = (16, 32, 96, 256)):
        super().__init__()
        self.conv_in = HipConv3x3(conditioning_channels, block_out_channels[0])
        blocks = []
        for i in range(len(block_out_channels) - 1):
            blocks.append(HipConv3x3(block_out_channels[i], block_out_channels[i]))
            blocks.append(HipConv3x3(block_out_channels[i], block_out_channels[i + 1], stride=2))
        self.blocks = nn.ModuleList(blocks)
        self.conv_out = HipConv3x3(block_out_channels[-1], conditioning_embedding_channels)
        nn.init.zeros_(self.conv_out.weight)  # zero_module in diffusers; checkpoints overwrite it

    def pack(self, arena, dtype):
        self.conv_in.pack(arena, dtype)
        for b in self.blocks:
            b.pack(arena, dtype)
        self.conv_out.pack(arena, dtype)

    def forward(self, cond_nhwc: torch.Tensor) -> torch.Tensor:
        e = self.conv_in.run(cond_nhwc, act=K.ACT_SILU)
        for b in self.blocks:
            e = b.run(e, act=K.ACT_SILU)
        return self.conv_out.run(e)


class ControlNetModel(HipModelMixin, nn.Module):
    def __init__(self, in_channels: int = 4, conditioning_channels: int = 3, flip_sin_to_cos: bool = True, freq_shift: int = 0,
                 down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
                 only_cross_attention: bool = False, block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280),
                 layers_per_block: int = 2, downsample_padding: int = 1, mid_block_scale_factor: float = 1, act_fn: str = "silu",
                 norm_num_groups: int = 32, norm_eps: float = 1e-5, cross_attention_dim: int = 768,
                 attention_head_dim: Union[int, Tuple[int, ...]] = 8, use_linear_projection: bool = False,
                 conditioning_embedding_out_channels: Tuple[int, ...] = (16, 32, 96, 256), global_pool_conditions: bool = False,
                 **unused):
        super().__init__()
        self._init_exec()
        if use_linear_projection or global_pool_conditions:
            raise NotImplementedError("not part of the SD1.5 ControlNet configuration")
        cfg = dict(locals())
        for k in ("self", "unused", "__class__"):
            cfg.pop(k, None)
        self.config = FrozenConfig(cfg)
        time_embed_dim = block_out_channels[0] * 4
        self.conv_in = HipConv3x3(in_channels, block_out_channels[0])
        self.time_proj = Timesteps(block_out_channels[0], flip_sin_to_cos, freq_shift)
        self.time_embedding = TimestepEmbedding(block_out_channels[0], time_embed_dim, act_fn=act_fn)
        self.controlnet_cond_embedding = ControlNetConditioningEmbedding(block_out_channels[0], conditioning_channels,
                                                                         conditioning_embedding_out_channels)
        self.down_blocks = nn.ModuleList([])
        self.controlnet_down_blocks = nn.ModuleList([HipConv1x1(block_out_channels[0], block_out_channels[0])])
        if isinstance(attention_head_dim, int):
            attention_head_dim = (attention_head_dim,) * len(down_block_types)
        common = dict(temb_channels=time_embed_dim, resnet_eps=norm_eps, resnet_groups=norm_num_groups,
                      cross_attention_dim=cross_attention_dim, use_inflated_groupnorm=True, use_motion_module=False)
        output_channel = block_out_channels[0]
        for i, t in enumerate(down_block_types):
            input_channel, output_channel = output_channel, block_out_channels[i]
            is_final = i == len(block_out_channels) - 1
            self.down_blocks.append(get_down_block(t, num_layers=layers_per_block, in_channels=input_channel,
                                                   out_channels=output_channel, add_downsample=not is_final,
                                                   attn_num_head_channels=attention_head_dim[i], **common))
            for _ in range(layers_per_block + (0 if is_final else 1)):
                self.controlnet_down_blocks.append(HipConv1x1(output_channel, output_channel))
        self.controlnet_mid_block = HipConv1x1(block_out_channels[-1], block_out_channels[-1])
        self.mid_block = UNetMidBlock3DCrossAttn(in_channels=block_out_channels[-1], output_scale_factor=mid_block_scale_factor,
                                                 attn_num_head_channels=attention_head_dim[-1], **common)
        for m in list(self.controlnet_down_blocks) + [self.controlnet_mid_block]:
            nn.init.zeros_(m.weight)  # zero convs; checkpoints overwrite them
        # hint embeddings, one slot per control-image tensor object (most recently used last; see HipModelMixin._slots)
        self._hints: "OrderedDict[int, dict]" = OrderedDict()

    @classmethod
    def from_config(cls, config: dict, **kwargs):
        merged = {k: v for k, v in dict(config).items() if not k.startswith("_")}
        merged.update(kwargs)
        return cls(**merged)

    # ---------------------------------------------------------------------------------------
    def hint_slot(self, controlnet_cond: Optional[torch.Tensor] = None) -> Optional[dict]:
        """The slot of control-image tensor `controlnet_cond` ({"src", "version", "emb", "doubled"}) or the most recently used one."""
        if controlnet_cond is None:
            return next(reversed(self._hints.values())) if self._hints else None
        ent = self._hints.get(id(controlnet_cond))
        return ent if ent is not None and ent["src"] is controlnet_cond else None

    @property
    def _hint_emb(self):  # (most recently used slot: single-pipeline callers, tests)
        ent = self.hint_slot()
        return None if ent is None else ent["emb"]

    @property
    def _hint_key(self):
        ent = self.hint_slot()
        return None if ent is None else (ent["src"], ent["version"])

    @property
    def _hint_doubled(self) -> bool:
        ent = self.hint_slot()
        return bool(ent is not None and ent["doubled"])

    def hint_embedding(self, controlnet_cond: torch.Tensor, device) -> torch.Tensor:
        """controlnet_cond [B,3,H,W] in [0,1] -> NHWC embedding [B,H/8,W/8,C0]; cached while the same
        (unmodified) tensor object is passed, i.e. for all denoising steps of a window."""
        return self._hint(controlnet_cond, device)["emb"]

    def _hint(self, controlnet_cond: torch.Tensor, device) -> dict:
        ent = self.hint_slot(controlnet_cond)
        if ent is not None:
            self._hints.move_to_end(id(controlnet_cond))
            if ent["version"] != controlnet_cond._version:  # the same tensor with new contents (the next window's frames): in place
                emb, doubled = self._embed_hints(controlnet_cond, device)
                ent["emb"].copy_(emb)
                ent["doubled"], ent["version"] = doubled, controlnet_cond._version
            return ent
        emb, doubled = self._embed_hints(controlnet_cond, device)
        ent = {"src": controlnet_cond, "version": controlnet_cond._version, "emb": emb, "doubled": doubled}
        self._hints.pop(id(controlnet_cond), None)
        self._hints[id(controlnet_cond)] = ent
        while len(self._hints) > self.MAX_CACHE_SLOTS:
            self._hints.popitem(last=False)
        return ent

    def _embed_hints(self, controlnet_cond: torch.Tensor, device):
        """The eight-convolution hint embedding (full-resolution 16- and 32-channel layers: the most expensive per-window
        work).  Hints that `prep_control_images` doubled for classifier-free guidance (`torch.cat([ctrl] * 2)`, reference
        :268-269) carry `_cfg_doubled`: the two halves are the same images, so one half is embedded and repeated -- the same
        bits, half the work."""
        ce = self.controlnet_cond_embedding
        cond = controlnet_cond
        doubled = bool(getattr(cond, "_cfg_doubled", False)) and cond.shape[0] % 2 == 0
        if doubled:
            half = cond.shape[0] // 2
            # the mark is a hint, the tensor is the truth: a caller may have edited one half in place since (regional control)
            doubled = bool(torch.equal(cond[:half], cond[half:]))
        # (`doubled` is what forward_body may rely on for THIS embedding: the shared prefix / the one-problem form of the CFG halves)
        if doubled:
            cond = cond[: cond.shape[0] // 2]
        emb = ce(K.ncfhw_to_nhwc(cond.to(device).unsqueeze(2), ce.conv_in.cin_pad, self.act_dtype))
        return (K.repeat_batch(emb) if doubled else emb), doubled

    def refresh_window_caches(self, src: Optional[torch.Tensor] = None, hint_src: Optional[torch.Tensor] = None) -> int:
        """HipModelMixin.refresh_window_caches + the hint embedding of the control images `hint_src` (default: the most recently used), in place."""
        n = super().refresh_window_caches(src)
        ent = self.hint_slot(hint_src)
        if ent is not None:
            emb, doubled = self._embed_hints(ent["src"], ent["emb"].device)
            ent["emb"].copy_(emb)
            ent["doubled"], ent["version"] = doubled, ent["src"]._version
            n += 1
        return n

    def prepare(self, device=None, dtype=None):
        self._hints.clear()
        return super().prepare(device, dtype)

    def residual_scales(self, conditioning_scale: float, guess_mode: bool) -> List[float]:
        n = len(self.controlnet_down_blocks) + 1
        if guess_mode:
            return [float(s) * conditioning_scale for s in torch.logspace(-1, 0, n)]
        return [float(conditioning_scale)] * n

    @torch.no_grad()
    def forward_body(self, x: torch.Tensor, timestep, encoder_hidden_states: torch.Tensor, controlnet_cond: torch.Tensor,
                     conditioning_scale: float = 1.0, guess_mode: bool = False, cfg_identical_halves: bool = False):
        """Everything up to the zero convolutions: -> (the 12 block outputs, the mid-block output, the 13 residual scales, `twice`);
        twice = the outputs hold one of two identical CFG halves (see below) and apply_zero_convs writes the residuals for both."""
        device = x.device
        self._ensure_ready(device)
        images = x.shape[0]
        ehs, cache = self._prompt(encoder_hidden_states, device)
        nb = ehs.shape[0]
        temb = self._time_embedding(timestep, 1, device)
        ctx = ExecCtx(b=images, f=1, dtype=self.act_dtype, temb=temb, emb_groups=1, ehs=ehs, frames_per_kv=1, kv_mod=nb,
                      gn_frames_per_stat=1, cache=cache)
        hint_ent = self._hint(controlnet_cond, device)
        hint = hint_ent["emb"]
        if hint.shape[0] != images:
            raise ValueError(f"controlnet_cond batch {hint.shape[0]} != sample batch {images}")
        # CFG-doubled input (the caller repeated one latent tensor, `cfg_identical_halves`) with CFG-doubled hints: the two
        # halves are identical up to the first cross-attention -- see UNet3DConditionModel.forward_nhwc
        first = self.down_blocks[0]
        from .context import dispatch
        same_inputs = (cfg_identical_halves and images % 2 == 0 and bool(hint_ent["doubled"]) and
                       (not torch.is_tensor(timestep) or timestep.numel() == 1))
        # The reference tiles the ControlNet's prompt as torch.cat([embeds] * frame_count) (modules/controlresiduals_pipeline.py:292,
        # SURVEY App. C-1): image z of the (b f) batch reads embeds[z % nb].  With an even number of frames per CFG half
        # ((images / 2) % nb == 0) image z and image z + images / 2 therefore read the SAME prompt row -- and under classifier-free
        # guidance they also hold the same latents (reference :797), the same control frame (:268-269) and the same timestep: the two
        # halves of the ControlNet's batch are the same problem from conv_in to the mid block.  It is solved once (half the batch, half of
        # the ControlNet's work); the zero convolutions write its residuals for both halves (apply_zero_convs, `twice`).
        if same_inputs and dispatch.cn_cfg_dedup and (images // 2) % nb == 0:
            half = images // 2
            hctx = dataclasses.replace(ctx, b=half)
            x = self.conv_in.run(x[:half], residual=hint[:half])
            outs = [x]
            for blk in self.down_blocks:
                x, o = blk(x, hctx)
                outs += o
            x = self.mid_block(x, hctx)
            return outs, x, self.residual_scales(conditioning_scale, guess_mode), True
        shared = same_inputs and dispatch.cfg_shared and getattr(first, "has_cross_attention", False)
        if shared:
            half = images // 2
            xh = self.conv_in.run(x[:half], residual=hint[:half])
            outs = [K.repeat_batch(xh)]  # (= torch.cat([xh, xh]), one read)
            x, o = first(xh, ctx, half_ctx=dataclasses.replace(ctx, b=half))
            outs += o
        else:
            x = self.conv_in.run(x, residual=hint)
            outs = [x]
        for blk in (self.down_blocks[1:] if shared else self.down_blocks):
            x, o = blk(x, ctx)
            outs += o
        x = self.mid_block(x, ctx)
        return outs, x, self.residual_scales(conditioning_scale, guess_mode), False

    @torch.no_grad()
    def apply_zero_convs(self, outs, x, scales, accumulate: Optional[Tuple[List[torch.Tensor], torch.Tensor]] = None, twice: bool = False):
        """The 13 zero convolutions: residual_i = scale_i * (zc_i(out_i)) [+ accumulate_i].  `accumulate` holds the running sums
        of the previous nets -- or, for the first net of a fused step, the UNet's own skip tensors and mid-block output
        (reference unet.py:567-576, 584-585: `sample + residual`): the add then happens in this GEMM's epilogue and the
        result IS the tensor the UNet's up blocks consume.
        twice: `outs` / `x` hold ONE of the two identical CFG halves (forward_body's de-duplicated batch); the residuals are written
        for both -- each half its own launch, on top of its own half of `accumulate`."""
        def one(zc, o, scale, prev):
            B_, h, w, c = o.shape
            rows = B_ * h * w
            a = o.view(rows, c)
            if not twice:
                return zc.run(a, alpha=scale, residual=None if prev is None else prev.view(rows, c)).view(B_, h, w, c)
            full = torch.empty((2 * B_, h, w, c), device=o.device, dtype=o.dtype)
            for hf in range(2):
                zc.run(a, alpha=scale, residual=None if prev is None else prev.view(2 * rows, c)[hf * rows:(hf + 1) * rows],
                       out=full.view(2 * rows, c)[hf * rows:(hf + 1) * rows])
            return full
        down = [one(zc, o, scales[i], None if accumulate is None else accumulate[0][i])
                for i, (zc, o) in enumerate(zip(self.controlnet_down_blocks, outs))]
        mid = one(self.controlnet_mid_block, x, scales[-1], None if accumulate is None else accumulate[1])
        return down, mid

    @torch.no_grad()
    def forward_nhwc(self, x: torch.Tensor, timestep, encoder_hidden_states: torch.Tensor, controlnet_cond: torch.Tensor,
                     conditioning_scale: float = 1.0, guess_mode: bool = False,
                     accumulate: Optional[Tuple[List[torch.Tensor], torch.Tensor]] = None, cfg_identical_halves: bool = False):
        """x: [B,h,w,cin_pad] activation dtype. Returns (12 NHWC residuals, mid), already scaled and --
        if `accumulate` holds the running sums of previous nets -- added to them."""
        outs, xm, scales, twice = self.forward_body(x, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale, guess_mode,
                                                    cfg_identical_halves)
        return self.apply_zero_convs(outs, xm, scales, accumulate, twice)

    def forward(self, sample: torch.Tensor, timestep, encoder_hidden_states: torch.Tensor, controlnet_cond: torch.Tensor,
                conditioning_scale: float = 1.0, class_labels=None, timestep_cond=None, attention_mask=None,
                cross_attention_kwargs=None, guess_mode: bool = False, return_dict: bool = False):
        """diffusers-style entry: sample [B,4,h,w] -> (tuple of 12 [B,C,h,w] views, [B,1280,h/8,w/8])."""
        if not sample.is_cuda:
            raise RuntimeError("ControlNetModel runs on the HIP device only (no CPU fallback)")
        self._ensure_ready(sample.device)
        x = K.ncfhw_to_nhwc(sample.unsqueeze(2), self.conv_in.cin_pad, self.act_dtype)
        down, mid = self.forward_nhwc(x, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale, guess_mode)
        return tuple(d.permute(0, 3, 1, 2) for d in down), mid.permute(0, 3, 1, 2)


class MultiControlNetModel(nn.Module):
    """diffusers' MultiControlNetModel: runs every net and sums the residuals."""

    def __init__(self, controlnets: Sequence[ControlNetModel]):
        super().__init__()
        self.nets = nn.ModuleList(controlnets)

    @property
    def dtype(self):
        return self.nets[0].dtype

    def half(self):
        for n in self.nets:
            n.half()
        return self

    def prepare(self, device=None, dtype=None):
        for n in self.nets:
            n.prepare(device, dtype)
        return self

    def forward_bodies(self, x, timestep, encoder_hidden_states, controlnet_cond: Sequence[torch.Tensor],
                       conditioning_scale: Sequence[float], guess_mode: bool = False, cfg_identical_halves: bool = False, streams=None):
        """Every net up to its zero convolutions (see ControlNetModel.forward_body): the part of the stack that does not need
        the UNet's skip tensors.  streams: HIP streams to spread the nets over, round-robin (net i on streams[i % len]) -- the bodies are
        independent of each other; the caller orders the streams before (inputs) and after (the zero convolutions read every body)."""
        bodies = []
        for i, (net, cond, scale) in enumerate(zip(self.nets, controlnet_cond, conditioning_scale)):
            if streams:
                with torch.cuda.stream(streams[i % len(streams)]):
                    bodies.append(net.forward_body(x, timestep, encoder_hidden_states, cond, scale, guess_mode, cfg_identical_halves))
            else:
                bodies.append(net.forward_body(x, timestep, encoder_hidden_states, cond, scale, guess_mode, cfg_identical_halves))
        return bodies

    def finish(self, bodies, base: Optional[Tuple[List[torch.Tensor], torch.Tensor]] = None):
        """Zero convolutions of every net, summed -- on top of `base` = (the UNet's skips, its mid-block output) when given."""
        acc = base
        for net, (outs, xm, scales, twice) in zip(self.nets, bodies):
            acc = net.apply_zero_convs(outs, xm, scales, acc, twice)
        return acc

    def forward_nhwc(self, x, timestep, encoder_hidden_states, controlnet_cond: Sequence[torch.Tensor],
                     conditioning_scale: Sequence[float], guess_mode: bool = False, cfg_identical_halves: bool = False):
        acc = None
        for net, cond, scale in zip(self.nets, controlnet_cond, conditioning_scale):
            acc = net.forward_nhwc(x, timestep, encoder_hidden_states, cond, scale, guess_mode, accumulate=acc,
                                   cfg_identical_halves=cfg_identical_halves)
        return acc

    def forward(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale, class_labels=None,
                timestep_cond=None, attention_mask=None, cross_attention_kwargs=None, guess_mode=False, return_dict=False):
        net0 = self.nets[0]
        net0._ensure_ready(sample.device)
        x = K.ncfhw_to_nhwc(sample.unsqueeze(2), net0.conv_in.cin_pad, net0.act_dtype)
        down, mid = self.forward_nhwc(x, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale, guess_mode)
        return tuple(d.permute(0, 3, 1, 2) for d in down), mid.permute(0, 3, 1, 2)
