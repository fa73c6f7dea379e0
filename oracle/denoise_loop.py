"""fp32 oracle of the reference's denoising loop. Test infrastructure only.

Follows animatediff/pipelines/controlanimation_pipeline.py:
  :675            do_classifier_free_guidance = guidance_scale > 1.0
  :698-710        IP-Adapter: 4 image tokens (or zeros when there is no previous window) appended
  :720-722        lcm_prompt_embeds = cond only; prompt_embeds = cat([neg, pos]) under CFG
  :731-740        timesteps (custom LCM: set_timesteps(strength, steps, 50); else set_timesteps(steps)
                  and the get_timesteps slice when strength < 1, :615-622)
  :769-771        w_embedding from the RAW guidance_scale
  :790-855        the loop: ControlNet input selection (:811-813), UNet call, CFG combine, step
and modules/controlresiduals_pipeline.py:268-269,278-316 for the ControlNet stack.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch

from . import schedulers as S
from .controlnet import ControlNetConfig, multi_controlnet_residuals
from .unet3d import UNet3DConfig, unet3d_forward


@dataclass
class LoopInputs:
    latents: torch.Tensor                      # [1,4,f,h,w] initial latents (already noised / scaled)
    prompt_embeds: torch.Tensor                # [1,L,768]
    negative_prompt_embeds: torch.Tensor       # [1,L,768]
    guidance_scale: float
    num_inference_steps: int
    scheduler: str                             # "DDIMScheduler" | "LCMScheduler" | "EulerDiscreteScheduler" | "custom_lcm"
    scheduler_kwargs: dict = field(default_factory=lambda: dict(beta_start=0.00085, beta_end=0.012, beta_schedule="linear"))
    strength: float = 1.0
    use_lcm: bool = False                      # native LCM UNet + in-tree scheduler
    guess_mode: bool = False
    control_images: Optional[Sequence[torch.Tensor]] = None   # per net: [f,3,H,W] in [0,1]
    cond_scale: Optional[Sequence[float]] = None
    ip_tokens: Optional[torch.Tensor] = None        # [1,4,768] cond image tokens (None + use_ip -> zeros)
    ip_uncond_tokens: Optional[torch.Tensor] = None
    use_ip: bool = False
    step_noises: Optional[Sequence[torch.Tensor]] = None  # one CPU noise draw per step for LCM samplers


def make_scheduler(name: str, kwargs: dict):
    if name == "custom_lcm":
        return S.CustomLCM()
    if name == "DDIMScheduler":
        return S.DDIM(**kwargs)
    if name == "LCMScheduler":
        return S.DiffusersLCM(**kwargs)
    if name == "EulerDiscreteScheduler":
        return S.EulerDiscrete(**kwargs)
    extra = {"EulerAncestralDiscreteScheduler": S.EulerAncestral, "LMSDiscreteScheduler": S.LMSDiscrete,
             "DPMSolverMultistepScheduler": S.DPMSolverMultistep, "PNDMScheduler": S.PNDM}
    if name in extra:
        return extra[name](**kwargs)
    raise NotImplementedError(name)


def denoise_loop(unet_sd, unet_cfg: UNet3DConfig, inp: LoopInputs, controlnets: Optional[Sequence[dict]] = None,
                 cn_cfg: Optional[ControlNetConfig] = None, ip: Optional[dict] = None) -> Dict[str, object]:
    do_cfg = inp.guidance_scale > 1.0
    pos, neg = inp.prompt_embeds.float(), inp.negative_prompt_embeds.float()
    if inp.use_ip:
        tok = inp.ip_tokens if inp.ip_tokens is not None else torch.zeros(1, 4, pos.shape[-1])
        utok = inp.ip_uncond_tokens if inp.ip_uncond_tokens is not None else torch.zeros(1, 4, pos.shape[-1])
        pos = torch.cat([pos, tok.float()], dim=1)
        neg = torch.cat([neg, utok.float()], dim=1)
    lcm_prompt = pos
    prompt = torch.cat([neg, pos]) if do_cfg else pos

    sched = make_scheduler("custom_lcm" if inp.use_lcm else inp.scheduler, inp.scheduler_kwargs)
    if inp.use_lcm:
        sched.set_timesteps(inp.strength, inp.num_inference_steps, 50)
        timesteps = sched.timesteps
    else:
        sched.set_timesteps(inp.num_inference_steps)
        timesteps = sched.timesteps
        if inp.strength < 1:
            init = min(int(inp.num_inference_steps * inp.strength), inp.num_inference_steps)
            timesteps = timesteps[max(inp.num_inference_steps - init, 0):]

    latents = inp.latents.float().clone()
    f = latents.shape[2]
    w_emb = S.get_w_embedding(torch.tensor([inp.guidance_scale]), 256)
    prep = None
    if controlnets:
        prep = [img.float() for img in inp.control_images]
        if do_cfg and not inp.guess_mode and not inp.use_lcm:
            prep = [torch.cat([p] * 2) for p in prep]  # controlresiduals_pipeline.py:268-269
    eps_hist: List[torch.Tensor] = []
    raw_hist: List[torch.Tensor] = []   # the UNet's raw output (both CFG halves)
    lat_hist: List[torch.Tensor] = []
    denoised = None
    for i, t in enumerate(timesteps):
        model_in = torch.cat([latents] * 2) if do_cfg else latents
        model_in = sched.scale_model_input(model_in, t)
        lcm_in = sched.scale_model_input(latents, t)
        down = mid = None
        if controlnets:
            single = inp.use_lcm or inp.guess_mode or not do_cfg
            # the reference casts the ControlNet's latent input and prompt to fp16 (controlresiduals_pipeline.py:295-297)
            cn_in = (lcm_in if single else model_in).half().float()
            cn_prompt = (lcm_prompt if single else prompt).half().float()
            down, mid = multi_controlnet_residuals(
                controlnets, cn_cfg, cn_in, t, cn_prompt,
                frame_count=f, prep_images=prep, cond_scale=inp.cond_scale, guess_mode=inp.guess_mode,
                strip_tokens=4 if inp.use_ip else 0)
        noise = inp.step_noises[i] if inp.step_noises is not None else None
        if inp.use_lcm:
            pred = unet3d_forward(unet_sd, unet_cfg, lcm_in, t, lcm_prompt, down, mid, timestep_cond=w_emb, ip=ip)
            raw_hist.append(pred)
            eps_hist.append(pred)
            latents, denoised = sched.step(pred, i, t, latents, noise=noise)
        else:
            pred = unet3d_forward(unet_sd, unet_cfg, model_in, t, prompt, down, mid, ip=ip)
            raw_hist.append(pred)
            if do_cfg:
                pu, pc = pred.chunk(2)
                pred = pu + inp.guidance_scale * (pc - pu)
            eps_hist.append(pred)
            if isinstance(sched, (S.DiffusersLCM, S.EulerAncestral)):
                latents, _ = sched.step(pred, t, latents, noise=noise)
            else:
                latents, _ = sched.step(pred, t, latents)
        lat_hist.append(latents)
    final = denoised if inp.use_lcm else latents  # pipeline decodes `denoised` for native LCM (:859-863)
    return {"timesteps": timesteps, "eps": eps_hist, "eps_raw": raw_hist, "latents": lat_hist, "final": final}
