"""Per-forward execution context shared by the blocks of one UNet3D / ControlNet pass."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import torch


@dataclass
class ExecCtx:
    b: int                      # batch elements (CFG halves)
    f: int                      # frames per batch element; images = b * f
    dtype: torch.dtype          # activation dtype (bf16 / fp16)
    temb: torch.Tensor          # fp32 [emb_groups, sum of resnet channels]: every time_emb_proj at once
    emb_groups: int             # rows of `temb` (b, or 1 when one timestep serves all images)
    ehs: Optional[torch.Tensor] = None   # [nb, L, cross_dim] prompt embeddings in `dtype`
    frames_per_kv: int = 1      # images sharing one prompt row block (f for the UNet)
    kv_mod: int = 0             # reference ControlNet prompt tiling (see ca_attention kv_mod)
    gn_frames_per_stat: int = 1  # 1 = per-frame GroupNorm (v2 / ControlNet), f = cross-frame (v1)
    cache: dict = field(default_factory=dict)   # text K/V reused across denoising steps

    @property
    def images(self) -> int:
        return self.b * self.f

    def rows_per_emb_group(self, h: int, w: int) -> int:
        return self.images // self.emb_groups * h * w
