"""ControlAnimatePipeline facade: what scripts/vid2vid.py instantiates and calls per window.

Call surface of the reference's modules/controlanimate_pipeline.py:
    ControlAnimatePipeline(config)                                                  (:25-121)
    .animate(input_frames, last_output_frames, config, image_prompt_embeds=None,
             uncond_image_prompt_embeds=None) -> frames                             (:124-170)
and the config keys it reads (inference_config_path / motion_module / use_lcm / controlnets /
cond_scale / scheduler / use_ipadapter / seed / width / height / steps / strength / guidance_scale /
frame_count / overlaps / epoch / guess_mode / ipa_scale / use_img2img ...).

`ControlAnimatePipeline(config)` builds everything from the LOCAL paths the config names, step for step as the reference
constructor does (tokenizer / text encoder / VAE / UNet from `pretrained_model_path/<subfolder>` or `vae_path` /
`pretrained_lcm_model_path`, ControlNets by name from local directories or the Hugging Face cache, scheduler by name with
the `noise_scheduler_kwargs` of `inference_config_path`, IP-Adapter from models/IP-Adapter, `load_weights` for the motion
module / DreamBooth / LoRAs, the easynegative textual inversion when its file exists, `maybe_convert_prompt`).  There is
no network: a model that is not on disk is an error naming the places that were searched.

Alternatively models are injected:
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


IP_IMAGE_ENCODER_PATH = "models/IP-Adapter/models/image_encoder/"   # modules/controlanimate_pipeline.py:80
IP_CKPT_PATH = "models/IP-Adapter/models/ip-adapter_sd15.bin"       # :81
TI_PATH = "models/TI/easynegative.safetensors"                       # :118


def components_from_config(config) -> Dict[str, Any]:
    """The model-building half of the reference constructor (modules/controlanimate_pipeline.py:27-48, 77-84) on local files."""
    import yaml
    from . import local_models as LM
    from .unet import UNet3DConditionModel
    with open(_get(config, "inference_config_path")) as fh:
        inference_config = yaml.safe_load(fh)
    use_lcm = bool(_get(config, "use_lcm", 0))
    base = _get(config, "pretrained_model_path")
    comp: Dict[str, Any] = dict(noise_scheduler_kwargs=inference_config.get("noise_scheduler_kwargs"), from_config=True)
    comp["tokenizer"] = LM.load_tokenizer(base)
    comp["text_encoder"] = LM.load_text_encoder(base)
    comp["vae"] = LM.load_vae(base, _get(config, "vae_path", "") or "")
    if not use_lcm:
        comp["unet"] = UNet3DConditionModel.from_pretrained_2d(base, subfolder="unet", use_safetensors=_has_safetensors(base, "unet"),
                                                               unet_additional_kwargs=inference_config["unet_additional_kwargs"])
    else:
        comp["unet"] = UNet3DConditionModel.from_pretrained_2d(_get(config, "pretrained_lcm_model_path"), subfolder="unet", use_safetensors=True,
                                                               unet_additional_kwargs=inference_config["unet_additional_kwargs"])
    names = _get(config, "controlnets", None)
    comp["controlnets"] = [LM.load_controlnet(n) for n in names] if names else []
    if bool(_get(config, "use_ipadapter", 0)):
        comp["image_encoder"] = LM.load_image_encoder(IP_IMAGE_ENCODER_PATH)
        comp["ip_adapter_ckpt"] = LM.read_checkpoint(IP_CKPT_PATH)
    return comp


def _has_safetensors(base, sub) -> bool:
    import os
    from .local_models import resolve_model_dir
    return os.path.isfile(os.path.join(resolve_model_dir(base, sub), "diffusion_pytorch_model.safetensors"))


class ControlAnimatePipeline:
    def __init__(self, config, components: Optional[Dict[str, Any]] = None, device=None, skeleton: bool = False):
        """skeleton=True (ranks > 0 of vid2vid.run_video_sharded): the same object, built from the config files WITHOUT reading
        any weight file -- the packed weight arenas are then received from rank 0 (`weight_buffers`)."""
        if skeleton and components is None:
            from .local_models import skeleton_weights
            with skeleton_weights():
                self.__init__(config, None, device, skeleton=False)
            return
        self.use_lcm = bool(_get(config, "use_lcm", 0))
        # the reference hard-wires "cuda"; without a GPU the object can still be built (files read, weights fused, prompts
        # converted) but animate() fails loudly: the execution path has no CPU fallback
        self.device = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        built_here = components is None
        if built_here:
            components = components_from_config(config)
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
            # reference :133-135: Compel(tokenizer, text_encoder)(prompt) -- the configs' prompts use its weighting syntax
            from .prompt_weighting import Compel
            self.encode_prompt = Compel(tokenizer=self.pipeline.tokenizer, text_encoder=self.pipeline.text_encoder, device=self.device)
        self.use_ipadapter = bool(_get(config, "use_ipadapter", 0))
        if self.use_ipadapter:
            ip = IPAdapter(self.pipeline, components.get("image_encoder"), components.get("ip_adapter_ckpt"), self.device, num_tokens=4)
            self.pipeline.ip_adapter = ip
            if self.multicontrolnetresiduals_pipeline is not None:
                ip.set_ip_adapter_4controlanimate(self.multicontrolnetresiduals_pipeline)
        from .local_models import is_skeleton
        if built_here and not is_skeleton():  # reference :87-106: motion module, DreamBooth checkpoint, LoRAs, motion LoRAs
            from .weight_ingest import load_weights
            lora_paths = _get(config, "lora_model_paths", "") or ""
            kw = {} if self.use_lcm else dict(dreambooth_model_path=_get(config, "dreambooth_path", "") or "",
                                              lora_model_path=lora_paths, lora_alpha=_get(config, "lora_weights", [0.8]))
            load_weights(self.pipeline, motion_module_path=_get(config, "motion_module", "") or "",
                         motion_module_lora_configs=_get(config, "motion_module_lora_configs", []) or [], **kw)
        # the reference halves everything unless native LCM (:108-110,115); here fp16 is the activation
        # dtype of the packed weights in both cases, selected at prepare() time.
        self._dtype = components.get("dtype", torch.float16)
        self._prepared = False
        import os
        if built_here and os.path.isfile(TI_PATH):  # :118 (the reference fails without the file; it ships it in models/TI)
            self.pipeline.load_textual_inversion(TI_PATH, token="easynegative")
        if self.device.type == "cuda" and torch.cuda.is_available():
            self._prepare_models()
        self.prompt = _get(config, "prompt", "")
        self.n_prompt = _get(config, "n_prompt", "")
        if self.pipeline.tokenizer is not None:  # :120-121
            self.prompt = self.pipeline.maybe_convert_prompt(self.prompt, self.pipeline.tokenizer)
            self.n_prompt = self.pipeline.maybe_convert_prompt(self.n_prompt, self.pipeline.tokenizer)
        self._embeds = components.get("prompt_embeds"), components.get("negative_prompt_embeds")

    def twin(self) -> "ControlAnimatePipeline":
        """A second facade over the SAME models for a second window in flight on this GPU (chains.py): its own denoising pipeline
        (sampler, captured hipGraph, static buffers), its own residuals pipeline (control-image tensors, side streams) and generator;
        the models, the prompt encoder and the encoded prompts are shared.  Nothing is loaded or packed again."""
        from .chains import clone_pipeline, clone_residuals_pipeline
        if not self._prepared:
            self._prepare_models()
        other = object.__new__(ControlAnimatePipeline)
        other.__dict__.update(self.__dict__)
        other.pipeline = clone_pipeline(self.pipeline)
        other.multicontrolnetresiduals_pipeline = clone_residuals_pipeline(self.multicontrolnetresiduals_pipeline)
        self._prompt_embeds() if self.encode_prompt is not None or self._embeds[0] is not None else None
        other._embeds = self._embeds
        return other

    def _prepare_models(self):
        """Packs the weights into their device arenas (needs the HIP device; there is no CPU path)."""
        models = [self.pipeline.unet] + (list(self.multicontrolnetresiduals_pipeline.controlnets) if self.multicontrolnetresiduals_pipeline else [])
        models += [m for m in (self.pipeline.vae, self.pipeline.text_encoder) if hasattr(m, "prepare") and getattr(m, "arena", 0) is None]
        for m in models:
            m.prepare(self.device, self._dtype)
        self._prepared = True

    def weight_buffers(self) -> list:
        """Every device tensor the execution path reads that came out of a checkpoint: the packed weight arena of each
        model (ONE contiguous buffer each) and the few tables that are used unpacked (CLIP's token / position embeddings).
        What rank 0 broadcasts to the other ranks of a window-sharded run (window_shard.broadcast_weights), in a fixed order."""
        if not self._prepared:
            self._prepare_models()
        models = [self.pipeline.unet] + (list(self.multicontrolnetresiduals_pipeline.controlnets) if self.multicontrolnetresiduals_pipeline else [])
        models += [m for m in (self.pipeline.vae, self.pipeline.text_encoder) if m is not None and getattr(m, "arena", None) is not None]
        ip = getattr(self.pipeline, "ip_adapter", None)
        if ip is not None and getattr(getattr(ip, "image_encoder", None), "arena", None) is not None:
            models.append(ip.image_encoder)
        bufs = [m.arena.buffer for m in models]
        te = self.pipeline.text_encoder
        emb = getattr(getattr(te, "text_model", None), "embeddings", None)
        if emb is not None:
            bufs += [emb.token_embedding.weight.data, emb.position_embedding.weight.data]
        if ip is not None:
            bufs += [p_.data for p_ in getattr(ip, "image_proj_model", torch.nn.Module()).parameters()]
        return bufs

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
        if not self._prepared:
            self._prepare_models()
        seed = int(_get(config, "seed", 0))
        torch.manual_seed(seed)                                   # global RNG: in-tree LCM noise (:129)
        self.generator = torch.Generator(device="cpu").manual_seed(seed)  # initial latents (:130)
        pos, neg = self._prompt_embeds()
        if image_prompt_embeds is not None:  # a fixed IP-Adapter image prompt instead of the previous window's first frame
            extra = dict(extra, image_prompt_embeds=image_prompt_embeds, uncond_image_prompt_embeds=uncond_image_prompt_embeds)
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
