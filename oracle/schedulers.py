"""fp32 oracle of the sampler arithmetic on the path. Test infrastructure only.

CustomLCM  = the reference's in-tree LCMScheduler (native-LCM path), restated from
             animatediff/pipelines/controlanimation_pipeline.py:1375-1633 (live method bodies:
             __init__ :1375-1426, set_timesteps :1487-1510, scalings :1512-1518, step :1520-1609,
             add_noise :1612-1633).  Pinned by tests/golden/lcm_custom.npz (generated from the
             reference class itself).
DDIM / DiffusersLCM / EulerDiscrete = diffusers==0.23.0 classes selected by name at
             modules/controlanimate_pipeline.py:52-67 and built with noise_scheduler_kwargs
             (configs/inference/inference-v*.yaml:23-27).  Third-party, absent from /root/reference:
             restated from the published algorithm -- PARITY UNPINNED.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch


def make_betas(beta_start: float, beta_end: float, beta_schedule: str, n: int = 1000) -> torch.Tensor:
    if beta_schedule == "linear":
        return torch.linspace(beta_start, beta_end, n, dtype=torch.float32)
    if beta_schedule == "scaled_linear":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    raise NotImplementedError(beta_schedule)


class _Base:
    init_noise_sigma = 1.0
    order = 1

    def __init__(self, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", num_train_timesteps=1000):
        self.num_train_timesteps = num_train_timesteps
        self.betas = make_betas(beta_start, beta_end, beta_schedule, num_train_timesteps)
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0)
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1)
        self.num_inference_steps = None

    def scale_model_input(self, sample, t=None):
        return sample

    def add_noise(self, original, noise, timesteps):
        a = self.alphas_cumprod[timesteps].to(original.dtype)
        sa = (a ** 0.5).flatten()
        sb = ((1 - a) ** 0.5).flatten()
        while sa.dim() < original.dim():
            sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
        return sa * original + sb * noise


class CustomLCM(_Base):
    """Reference in-tree scheduler; constructed at controlanimation_pipeline.py:98-100 with
    beta_start=0.00085, beta_end=0.012, beta_schedule='scaled_linear', prediction_type='epsilon'."""

    def __init__(self, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear"):
        super().__init__(beta_start, beta_end, beta_schedule)

    def set_timesteps(self, strength: float, num_inference_steps: int, lcm_origin_steps: int = 50):
        self.num_inference_steps = num_inference_steps
        c = self.num_train_timesteps // lcm_origin_steps
        origin = np.asarray(list(range(1, int(lcm_origin_steps * strength) + 1))) * c - 1
        skipping = len(origin) // num_inference_steps
        self.timesteps = torch.from_numpy(origin[::-skipping][:num_inference_steps].copy())

    @staticmethod
    def scalings(t):
        sigma_data = 0.5
        c_skip = sigma_data ** 2 / ((t / 0.1) ** 2 + sigma_data ** 2)
        c_out = (t / 0.1) / ((t / 0.1) ** 2 + sigma_data ** 2) ** 0.5
        return c_skip, c_out

    def step(self, model_output, timeindex: int, timestep, sample, noise: Optional[torch.Tensor] = None):
        """Returns (prev_sample, denoised). `noise` stands for the torch.randn draw at :1601
        (global CPU RNG in the reference); None -> draw it here the same way."""
        prev_index = timeindex + 1
        prev_t = self.timesteps[prev_index] if prev_index < len(self.timesteps) else timestep
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t, b_prev = 1 - a_t, 1 - a_prev
        c_skip, c_out = self.scalings(timestep)
        pred_x0 = (sample - b_t.sqrt() * model_output) / a_t.sqrt()
        denoised = c_out * pred_x0 + c_skip * sample
        if len(self.timesteps) > 1:
            if noise is None:
                noise = torch.randn(model_output.shape)
            prev = a_prev.sqrt() * denoised + b_prev.sqrt() * noise
        else:
            prev = denoised
        return prev, denoised


class DiffusersLCM(_Base):
    """diffusers 0.23.0 LCMScheduler as the reference builds it for LCM-LoRA configs:
    LCMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule='linear')."""

    def __init__(self, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", original_inference_steps=50):
        super().__init__(beta_start, beta_end, beta_schedule)
        self.original_inference_steps = original_inference_steps
        self._step_index = None

    def set_timesteps(self, num_inference_steps: int):
        self.num_inference_steps = num_inference_steps
        c = self.num_train_timesteps // self.original_inference_steps
        origin = np.asarray(list(range(1, self.original_inference_steps + 1))) * c - 1
        skipping = len(origin) // num_inference_steps
        self.timesteps = torch.from_numpy(origin[::-skipping][:num_inference_steps].copy())
        self._step_index = None

    def step(self, model_output, timestep, sample, generator=None, noise: Optional[torch.Tensor] = None):
        if self._step_index is None:
            self._step_index = int((self.timesteps == timestep).nonzero()[0])
        prev_index = self._step_index + 1
        prev_t = self.timesteps[prev_index] if prev_index < len(self.timesteps) else timestep
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        c_skip, c_out = CustomLCM.scalings(timestep)
        pred_x0 = (sample - (1 - a_t).sqrt() * model_output) / a_t.sqrt()
        denoised = c_out * pred_x0 + c_skip * sample
        if len(self.timesteps) > 1:
            if noise is None:
                noise = torch.randn(model_output.shape, generator=generator)  # randn_tensor, CPU generator
            prev = a_prev.sqrt() * denoised + (1 - a_prev).sqrt() * noise
        else:
            prev = denoised
        self._step_index += 1
        return prev, denoised


class DDIM(_Base):
    """diffusers 0.23.0 DDIMScheduler defaults + noise_scheduler_kwargs: clip_sample=True,
    set_alpha_to_one=True, steps_offset=0, 'leading' spacing, eta=0."""

    def __init__(self, beta_start=0.00085, beta_end=0.012, beta_schedule="linear", clip_sample=True, clip_sample_range=1.0):
        super().__init__(beta_start, beta_end, beta_schedule)
        self.clip_sample, self.clip_sample_range = clip_sample, clip_sample_range

    def set_timesteps(self, num_inference_steps: int):
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts)

    def step(self, model_output, timestep, sample):
        prev_t = timestep - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        pred_x0 = (sample - (1 - a_t) ** 0.5 * model_output) / a_t ** 0.5
        if self.clip_sample:
            pred_x0 = pred_x0.clamp(-self.clip_sample_range, self.clip_sample_range)
        direction = (1 - a_prev) ** 0.5 * model_output  # eta = 0, use_clipped_model_output=False
        return a_prev ** 0.5 * pred_x0 + direction, pred_x0


class EulerDiscrete(_Base):
    """diffusers 0.23.0 EulerDiscreteScheduler defaults ('linspace' spacing, linear interpolation,
    epsilon prediction, s_churn=0)."""

    def __init__(self, beta_start=0.00085, beta_end=0.012, beta_schedule="linear"):
        super().__init__(beta_start, beta_end, beta_schedule)
        sig = ((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5
        self.sigmas = torch.cat([sig.flip(0), torch.zeros(1)])
        self._step_index = None

    @property
    def init_noise_sigma(self):
        return float(self.sigmas.max())  # 'linspace' spacing

    def set_timesteps(self, num_inference_steps: int):
        self.num_inference_steps = num_inference_steps
        ts = np.linspace(0, self.num_train_timesteps - 1, num_inference_steps, dtype=np.float32)[::-1].copy()
        sig = np.array(((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5)
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0.0]]).astype(np.float32))
        self.timesteps = torch.from_numpy(ts)
        self._step_index = None

    def _index(self, timestep):
        if self._step_index is None:
            self._step_index = int((self.timesteps == timestep).nonzero()[0])
        return self._step_index

    def scale_model_input(self, sample, t=None):
        s = self.sigmas[self._index(t)]
        return sample / ((s ** 2 + 1) ** 0.5)

    def step(self, model_output, timestep, sample):
        i = self._index(timestep)
        sigma = self.sigmas[i]
        pred_x0 = sample - sigma * model_output
        derivative = (sample - pred_x0) / sigma
        prev = sample + derivative * (self.sigmas[i + 1] - sigma)
        self._step_index += 1
        return prev, pred_x0


def get_w_embedding(w: torch.Tensor, embedding_dim: int = 512) -> torch.Tensor:
    """Guidance-scale embedding for the native-LCM UNet (controlanimation_pipeline.py:477-498);
    the pipeline passes the raw guidance_scale (:769-770)."""
    w = w.float() * 1000.0
    half = embedding_dim // 2
    emb = torch.log(torch.tensor(10000.0)) / (half - 1)
    emb = torch.exp(torch.arange(half, dtype=torch.float32) * -emb)
    emb = w[:, None] * emb[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)
    if embedding_dim % 2 == 1:
        emb = torch.nn.functional.pad(emb, (0, 1))
    return emb
