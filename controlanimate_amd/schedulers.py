"""Samplers of the denoising loop as host-side coefficient tables for ONE fused device kernel.

Every sampler the reference can select (modules/controlanimate_pipeline.py:52-67) is an affine
update in (sample, eps, noise) once the timestep is fixed; the loop therefore runs
`ca_cfg_scheduler_step` (CFG combine + update, include/controlanimate_hip.h) with 7 coefficients:

    x0   = (x - c0*eps) * c1          (optionally clamped)
    den  = c2*x0 + c3*x
    prev = c4*den + c5*eps + c6*noise

Implemented: the reference's in-tree native-LCM scheduler (`LCMScheduler` here, as in
animatediff/pipelines/controlanimation_pipeline.py:977,1375-1633) and the diffusers 0.23.0 classes
DDIMScheduler, LCMScheduler (exported as `DiffusersLCMScheduler`) and EulerDiscreteScheduler with
the constructor defaults the reference relies on.  The other names in the reference's table
(DPMSolverMultistep, EulerAncestral, LMS, PNDM) are multistep/ancestral samplers: not built yet
(`get_scheduler` raises NotImplementedError).  All scalar math is done in float64 on the host.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import numpy as np
import torch


def _betas(beta_start: float, beta_end: float, beta_schedule: str, n: int) -> np.ndarray:
    if beta_schedule == "linear":
        return torch.linspace(beta_start, beta_end, n, dtype=torch.float32).numpy().astype(np.float64)
    if beta_schedule == "scaled_linear":
        return (torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2).numpy().astype(np.float64)
    raise NotImplementedError(f"{beta_schedule} is not implemented")


class _SchedulerBase:
    order = 1
    needs_noise = False
    clip = 0.0

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02,
                 beta_schedule: str = "linear", **_):
        self.num_train_timesteps = num_train_timesteps
        betas32 = torch.from_numpy(_betas(beta_start, beta_end, beta_schedule, num_train_timesteps)).float()
        # cumprod in fp32 exactly like the reference / diffusers, then widened for the scalar math
        self.alphas_cumprod = torch.cumprod(1.0 - betas32, dim=0)
        self._ac = self.alphas_cumprod.double().numpy()
        self.init_noise_sigma = 1.0
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1)
        self.num_inference_steps: Optional[int] = None

    def input_scale(self, index: int) -> float:
        """scale_model_input as a scalar factor."""
        return 1.0

    def add_noise(self, original: torch.Tensor, noise: torch.Tensor, timesteps) -> torch.Tensor:
        t = int(torch.as_tensor(timesteps).reshape(-1)[0])
        a = float(self._ac[t])
        return math.sqrt(a) * original + math.sqrt(1.0 - a) * noise

    def coefficients(self, index: int) -> Tuple[List[float], float]:
        raise NotImplementedError


def _lcm_scalings(t: float) -> Tuple[float, float]:
    sigma_data = 0.5
    c_skip = sigma_data ** 2 / ((t / 0.1) ** 2 + sigma_data ** 2)
    c_out = (t / 0.1) / ((t / 0.1) ** 2 + sigma_data ** 2) ** 0.5
    return c_skip, c_out


class LCMScheduler(_SchedulerBase):
    """The reference's IN-TREE scheduler (native-LCM path): set_timesteps(strength, steps, origin)."""

    needs_noise = True

    def __init__(self, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", **kw):
        super().__init__(beta_start=beta_start, beta_end=beta_end, beta_schedule=beta_schedule, **kw)

    def set_timesteps(self, stength: float, num_inference_steps: int, lcm_origin_steps: int = 50, device=None):
        if num_inference_steps > self.num_train_timesteps:
            raise ValueError("num_inference_steps cannot exceed num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        c = self.num_train_timesteps // lcm_origin_steps
        origin = np.asarray(list(range(1, int(lcm_origin_steps * stength) + 1))) * c - 1
        skipping = len(origin) // num_inference_steps
        self.timesteps = torch.from_numpy(origin[::-skipping][:num_inference_steps].copy())

    def coefficients(self, index: int):
        ts = self.timesteps
        t = int(ts[index])
        prev_t = int(ts[index + 1]) if index + 1 < len(ts) else t
        a_t, a_prev = self._ac[t], (self._ac[prev_t] if prev_t >= 0 else 1.0)
        c_skip, c_out = _lcm_scalings(float(t))
        multi = len(ts) > 1
        return [math.sqrt(1 - a_t), 1.0 / math.sqrt(a_t), c_out, c_skip,
                math.sqrt(a_prev) if multi else 1.0, 0.0, math.sqrt(1 - a_prev) if multi else 0.0], self.clip


class DiffusersLCMScheduler(LCMScheduler):
    """diffusers 0.23.0 LCMScheduler: set_timesteps(num_inference_steps); same update rule."""

    def __init__(self, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", original_inference_steps=50, **kw):
        super().__init__(beta_start=beta_start, beta_end=beta_end, beta_schedule=beta_schedule, **kw)
        self.original_inference_steps = original_inference_steps

    def set_timesteps(self, num_inference_steps: int, device=None, original_inference_steps: Optional[int] = None):
        super().set_timesteps(1.0, num_inference_steps, original_inference_steps or self.original_inference_steps)


class DDIMScheduler(_SchedulerBase):
    def __init__(self, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", clip_sample=True, clip_sample_range=1.0,
                 set_alpha_to_one=True, steps_offset=0, **kw):
        super().__init__(beta_start=beta_start, beta_end=beta_end, beta_schedule=beta_schedule, **kw)
        self.clip = float(clip_sample_range) if clip_sample else 0.0
        self.final_alpha = 1.0 if set_alpha_to_one else float(self._ac[0])
        self.steps_offset = steps_offset

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        self.timesteps = torch.from_numpy(ts)

    def coefficients(self, index: int):
        t = int(self.timesteps[index])
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self._ac[t]
        a_prev = self._ac[prev_t] if prev_t >= 0 else self.final_alpha
        # eta = 0: prev = sqrt(a_prev) * clip(x0) + sqrt(1 - a_prev) * eps
        return [math.sqrt(1 - a_t), 1.0 / math.sqrt(a_t), 1.0, 0.0, math.sqrt(a_prev), math.sqrt(1 - a_prev), 0.0], self.clip


class EulerDiscreteScheduler(_SchedulerBase):
    def __init__(self, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", **kw):
        super().__init__(beta_start=beta_start, beta_end=beta_end, beta_schedule=beta_schedule, **kw)
        sig = np.sqrt((1 - self._ac) / self._ac)
        self.sigmas = np.concatenate([sig[::-1], [0.0]])
        self.init_noise_sigma = float(self.sigmas.max())  # 'linspace' timestep spacing

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ts = np.linspace(0, self.num_train_timesteps - 1, num_inference_steps, dtype=np.float32)[::-1].copy()
        sig = np.sqrt((1 - self._ac) / self._ac)
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        self.sigmas = np.concatenate([sig, [0.0]]).astype(np.float32).astype(np.float64)
        self.timesteps = torch.from_numpy(ts)
        self.init_noise_sigma = float(self.sigmas.max())

    def input_scale(self, index: int) -> float:
        return 1.0 / math.sqrt(self.sigmas[index] ** 2 + 1.0)

    def coefficients(self, index: int):
        # pred_x0 = x - sigma*eps; derivative = eps; prev = x + (sigma_next - sigma) * eps
        return [0.0, 1.0, 0.0, 1.0, 1.0, float(self.sigmas[index + 1] - self.sigmas[index]), 0.0], 0.0


SCHEDULERS = {
    "DDIMScheduler": DDIMScheduler,
    "LCMScheduler": DiffusersLCMScheduler,  # config name -> diffusers' class
    "EulerDiscreteScheduler": EulerDiscreteScheduler,
}


def get_scheduler(name: str, **kwargs):
    if name not in SCHEDULERS:
        raise NotImplementedError(f"scheduler {name!r} is not built yet (available: {sorted(SCHEDULERS)})")
    return SCHEDULERS[name](**kwargs)


def get_w_embedding(w: torch.Tensor, embedding_dim: int = 512, dtype=torch.float32) -> torch.Tensor:
    """Guidance-scale embedding of the native-LCM UNet (controlanimation_pipeline.py:477-498)."""
    assert len(w.shape) == 1
    w = w * 1000.0
    half = embedding_dim // 2
    emb = torch.log(torch.tensor(10000.0)) / (half - 1)
    emb = torch.exp(torch.arange(half, dtype=dtype) * -emb)
    emb = w.to(dtype)[:, None] * emb[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)
    if embedding_dim % 2 == 1:
        emb = torch.nn.functional.pad(emb, (0, 1))
    return emb
