"""Model configurations the reference ships (values of configs/inference/inference-v{1,2}.yaml and
of the SD1.5 / ControlNet config.json files its download scripts fetch), as plain dicts."""
from __future__ import annotations

import copy

SD15_UNET = dict(
    sample_size=64, in_channels=4, out_channels=4, center_input_sample=False, flip_sin_to_cos=True, freq_shift=0,
    down_block_types=("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"),
    up_block_types=("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"),
    block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, downsample_padding=1, mid_block_scale_factor=1,
    act_fn="silu", norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=768, attention_head_dim=8,
)

_MOTION = dict(num_attention_heads=8, num_transformer_block=1, attention_block_types=["Temporal_Self", "Temporal_Self"],
               temporal_position_encoding=True, temporal_attention_dim_div=1)

# configs/inference/inference-v1.yaml:1-21 (mm_sd_v14 / v15)
INFERENCE_V1 = dict(
    unet_use_cross_frame_attention=False, unet_use_temporal_attention=False, use_motion_module=True,
    motion_module_resolutions=[1, 2, 4, 8], motion_module_mid_block=False, motion_module_decoder_only=False,
    motion_module_type="Vanilla", motion_module_kwargs=dict(_MOTION, temporal_position_encoding_max_len=24),
)
# configs/inference/inference-v2.yaml:1-22 (mm_sd_v15_v2)
INFERENCE_V2 = dict(
    use_inflated_groupnorm=True, unet_use_cross_frame_attention=False, unet_use_temporal_attention=False,
    use_motion_module=True, motion_module_resolutions=[1, 2, 4, 8], motion_module_mid_block=True,
    motion_module_decoder_only=False, motion_module_type="Vanilla",
    motion_module_kwargs=dict(_MOTION, temporal_position_encoding_max_len=32),
)
NOISE_SCHEDULER_KWARGS = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="linear")

SD15_CONTROLNET = dict(
    in_channels=4, conditioning_channels=3, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
    cross_attention_dim=768, attention_head_dim=8, norm_num_groups=32, norm_eps=1e-5,
    conditioning_embedding_out_channels=(16, 32, 96, 256),
)


def unet_config(version: str = "v2", **overrides) -> dict:
    cfg = copy.deepcopy(SD15_UNET)
    cfg.update(copy.deepcopy(INFERENCE_V2 if version == "v2" else INFERENCE_V1))
    cfg.update(overrides)
    return cfg


def controlnet_config(**overrides) -> dict:
    cfg = copy.deepcopy(SD15_CONTROLNET)
    cfg.update(overrides)
    return cfg
