"""ControlAnimatePipeline facade: what scripts/vid2vid.py instantiates and calls per window.

Call surface of the reference's modules/controlanimate_pipeline.py:
    ControlAnimatePipeline(config)                                                  (:25-121)
    .animate(input_frames, last_output_frames, config, image_prompt_embeds=None,
             uncond_image_prompt_embeds=None) -> frames                             (:124-170)
and the config keys it reads (inference_config_path / motion_module / use_lcm / controlnets /
cond_scale / scheduler / use_ipadapter / seed / width / height / steps / strength / guidance_scale /
frame_count / overlaps / epoch / guess_mode / ipa_scale / use_img2img ...).

Models are injected (there is no network for Hugging Face downloads):
    components = dict(unet=UNet3DConditionModel, controlnets=[ControlNetModel...],
                      vae=None | controlanimate_amd.vae.AutoencoderKL | any diffusers-style VAE,
                      text_encoder=None | controlanimate_amd.clip.CLIPTextModel, tokenizer=None | CLIPTokenizer,
                      encode_prompt=None | callable(str)->Tensor[1,77,768]   (e.g. the reference's Compel object),
                      ip_adapter_ckpt=None|dict, image_encoder=None | clip.CLIPVisionModelWithProjection | callable)
With text_encoder + tokenizer and no encode_prompt the prompt is encoded unweighted (what Compel returns
for a prompt without weighting syntax; Compel itself, :133-135, is third-party host code and not rebuilt).
Checkpoint conversion / LoRA fusing: controlanimate_amd.weight_ingest.  With a VAE attached `animate`
returns PIL frames like the reference (:166, modules/utils.py:74-85); without one it returns the latents.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Optional

import torch

from .configs import NOISE_SCHEDULER_KWARGS
from .controlanimation_pipeline import ControlAnimationPipeline
from .controlresiduals_pipeline import MultiControlNetResidualsPipeline
from .ip_adapter import IPAdapter
from .schedulers import get_scheduler


def _get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default) if hasattr(cfg, key) else default


class ControlAnimatePipeline:
    def __init__(self, config, components: Dict[str, Any], device="cuda"):
        self.use_lcm = bool(_get(config, "use_lcm", 0))
        self.device = torch.device(device)
        unet = components["unet"]
        nets = components.get("controlnets") or []
        names = list(_get(config, "controlnets", None) or [f"controlnet-{i}" for i in range(len(nets))])
        self.multicontrolnetresiduals_pipeline = MultiControlNetResidualsPipeline(
            names, list(_get(config, "cond_scale", [1.0] * len(nets))), use_lcm=self.use_lcm, controlnets=nets,
            device=self.device, annotators=components.get("annotators")) if nets else None
        noise_kwargs = dict(components.get("noise_scheduler_kwargs") or NOISE_SCHEDULER_KWARGS)
        scheduler = None if self.use_lcm else get_scheduler(_get(config, "scheduler", "DDIMScheduler"), **noise_kwargs)
        self.pipeline = ControlAnimationPipeline(vae=components.get("vae"), text_encoder=components.get("text_encoder"),
                                                 tokenizer=components.get("tokenizer"), unet=unet, scheduler=scheduler).to(self.device)
        self.encode_prompt: Optional[Callable] = components.get("encode_prompt")
        if self.encode_prompt is None and self.pipeline.text_encoder is not None and self.pipeline.tokenizer is not None:
            self.encode_prompt = self._encode_plain
        self.use_ipadapter = bool(_get(config, "use_ipadapter", 0))
        if self.use_ipadapter:
            ip = IPAdapter(self.pipeline, components.get("image_encoder"), components.get("ip_adapter_ckpt"), self.device, num_tokens=4)
            self.pipeline.ip_adapter = ip
            if self.multicontrolnetresiduals_pipeline is not None:
                ip.set_ip_adapter_4controlanimate(self.multicontrolnetresiduals_pipeline)
        # the reference halves everything unless native LCM (:108-110,115); here fp16 is the activation
        # dtype of the packed weights in both cases, selected at prepare() time.
        unet.prepare(self.device, components.get("dtype", torch.float16))
        for n in nets:
            n.prepare(self.device, components.get("dtype", torch.float16))
        self.prompt = _get(config, "prompt", "")
        self.n_prompt = _get(config, "n_prompt", "")
        self._embeds = components.get("prompt_embeds"), components.get("negative_prompt_embeds")

    def _encode_plain(self, prompt: str) -> torch.Tensor:
        tok = self.pipeline.tokenizer
        ids = tok(prompt, padding="max_length", max_length=getattr(tok, "model_max_length", 77), truncation=True,
                  return_tensors="pt").input_ids
        return self.pipeline.text_encoder(ids.to(self.device))[0]

    def _prompt_embeds(self):
        if self._embeds[0] is not None:
            return self._embeds
        if self.encode_prompt is None:
            raise RuntimeError("no text encoder: pass components['encode_prompt'] or precomputed prompt_embeds")
        self._embeds = (self.encode_prompt(self.prompt), self.encode_prompt(self.n_prompt))
        return self._embeds

    def animate(self, input_frames, last_output_frames, config, image_prompt_embeds=None, uncond_image_prompt_embeds=None,
                **extra):
        seed = int(_get(config, "seed", 0))
        torch.manual_seed(seed)                                   # global RNG: in-tree LCM noise (:129)
        self.generator = torch.Generator(device="cpu").manual_seed(seed)  # initial latents (:130)
        pos, neg = self._prompt_embeds()
        out = self.pipeline(
            prompt_embeds=pos, negative_prompt_embeds=neg, input_frames=input_frames,
            num_inference_steps=int(_get(config, "steps")), strength=float(_get(config, "strength", 1.0)),
            guidance_scale=float(_get(config, "guidance_scale", 7.5)), width=int(_get(config, "width")),
            height=int(_get(config, "height")), video_length=int(_get(config, "frame_count")), generator=self.generator,
            overlaps=int(_get(config, "overlaps", 0)), multicontrolnetresiduals_pipeline=self.multicontrolnetresiduals_pipeline,
            epoch=_get(config, "epoch", 0), output_dir=_get(config, "output_video_dir", "tmp/output"),
            save_outputs=bool(_get(config, "save_frames", 0)), last_output_frames=last_output_frames, use_lcm=self.use_lcm,
            guess_mode=bool(_get(config, "guess_mode", 0)), ipa_scale=float(_get(config, "ipa_scale", 0.4)),
            use_img2img=bool(_get(config, "use_img2img", False)), **extra)
        videos = out.videos
        if self.pipeline.vae is None or extra.get("output_type") == "latent":
            return videos
        return frames_to_pil(videos)


def frames_to_pil(videos) -> List:
    """modules/utils.py:74-85 get_frames_pil_images: [b,c,t,h,w] in [0,1] -> list of (t b) PIL images
    (`(x * 255).astype(uint8)`: truncation, as the reference)."""
    import numpy as np
    from PIL import Image
    v = torch.as_tensor(videos).float().cpu()
    frames = v.permute(2, 0, 3, 4, 1).reshape(-1, v.shape[3], v.shape[4], v.shape[1])
    return [Image.fromarray((x * 255).numpy().astype(np.uint8).squeeze(-1) if x.shape[-1] == 1 else (x * 255).numpy().astype(np.uint8))
            for x in frames]
