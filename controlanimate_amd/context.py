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


class Dispatch:
    """Which of the one-launch / fused forms the host side hands to the library.  All on: that is the product, and no
    product module reads the environment to change it.  A/B timing runs flip these attributes from OUTSIDE the package
    (tools/ab_switches.py maps the CA_* variables of earlier rounds onto them for `bench.py`; tests set them directly)."""
    gemm_ar = True        # fragment-ordered W handed to ca_gemm (activation-resident K = 320 kernel)
    ff_fused = True       # ca_ff_fused for the C = 320 feed-forward
    tattn_fused = True    # ca_tattn_fused for the 64x64-latent motion modules
    xattn_fused = True    # ca_xattn_fused for the 64x64-latent text cross-attention
    attn_out_fused = True  # ... with the output projection + bias + residual as their last stage (ABI v12)
    xattn_ip_fused = True  # ... the IP-Adapter's image-prompt attention inside the text cross-attention's launch (ABI v13)
    conv_winograd = True  # Winograd F(2x2, 3x3) form of the deep convolutions at the small-latent levels (read at prepare() time)
    gn_winograd = True    # ... with the GroupNorm in front writing the transformed input itself (ca_groupnorm_args.wino_v)
    ln_row_sums = True    # LayerNorm statistics from the producing GEMM's epilogue
    repeat_kernel = True  # ca_repeat instead of torch.cat for the CFG-shared prefix
    ln_fold = True        # LayerNorm folded into the projection it feeds
    cfg_shared = True     # the prompt-independent prefix of the two CFG halves runs once
    cn_cfg_dedup = True   # ControlNet under non-guess CFG: the reference's prompt tiling makes its two batch halves the same problem -- solved once
    controlnet_streams = 2  # HIP streams the bodies of a stack of >= 3 ControlNets are spread over (independent up to the summed residuals)


dispatch = Dispatch()
