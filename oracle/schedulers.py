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


# ------------------------------------------------------------------------------------------------------------------
# The remaining samplers of the reference's table (modules/controlanimate_pipeline.py:52-61), restated from
# diffusers 0.23.0 in the SHAPE of its step functions (tensor arithmetic with history lists), i.e. independently of the
# coefficient tables of controlanimate_amd/schedulers.py which are tested against these.  Third-party arithmetic:
# parity unpinned (no diffusers here, no reference test vectors).
class EulerAncestral(EulerDiscrete):
    def step(self, model_output, timestep, sample, noise=None, generator=None):
        i = self._index(timestep)
        sigma = self.sigmas[i]
        pred_original = sample - sigma * model_output
        s_from, s_to = self.sigmas[i], self.sigmas[i + 1]
        s_up = (s_to ** 2 * (s_from ** 2 - s_to ** 2) / s_from ** 2) ** 0.5
        s_down = (s_to ** 2 - s_up ** 2) ** 0.5
        derivative = (sample - pred_original) / sigma
        prev = sample + derivative * (s_down - sigma)
        if noise is None:
            noise = torch.randn(model_output.shape, generator=generator)
        self._step_index += 1
        return prev + noise * s_up, pred_original


class LMSDiscrete(EulerDiscrete):
    def set_timesteps(self, num_inference_steps: int):
        super().set_timesteps(num_inference_steps)
        self.derivatives = []

    def _coeff(self, order, t, current_order):
        from scipy import integrate
        sig = [float(s) for s in self.sigmas]

        def f(tau):
            prod = 1.0
            for k in range(order):
                if current_order == k:
                    continue
                prod *= (tau - sig[t - k]) / (sig[t - current_order] - sig[t - k])
            return prod
        return integrate.quad(f, sig[t], sig[t + 1], epsrel=1e-4)[0]

    def step(self, model_output, timestep, sample, order: int = 4):
        i = self._index(timestep)
        sigma = self.sigmas[i]
        pred_original = sample - sigma * model_output
        self.derivatives.append((sample - pred_original) / sigma)
        if len(self.derivatives) > order:
            self.derivatives.pop(0)
        order = min(i + 1, order)
        coeffs = [self._coeff(order, i, k) for k in range(order)]
        prev = sample + sum(c * d for c, d in zip(coeffs, reversed(self.derivatives)))
        self._step_index += 1
        return prev, pred_original


class DPMSolverMultistep(_Base):
    """dpmsolver++ / order 2 / midpoint / lower_order_final, 'linspace' spacing, sigma formulation."""

    def __init__(self, beta_start=0.00085, beta_end=0.012, beta_schedule="linear", solver_order=2):
        super().__init__(beta_start, beta_end, beta_schedule)
        self.solver_order = solver_order

    def set_timesteps(self, num_inference_steps: int):
        self.num_inference_steps = num_inference_steps
        ts = np.linspace(0, self.num_train_timesteps - 1, num_inference_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
        sig = np.array(((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5)
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        last = float(((1 - self.alphas_cumprod[0]) / self.alphas_cumprod[0]) ** 0.5)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [last]]).astype(np.float32))
        self.timesteps = torch.from_numpy(ts)
        self.model_outputs = [None] * self.solver_order
        self.lower_order_nums = 0
        self._step = 0

    @staticmethod
    def _alpha_sigma(sigma):
        alpha = 1 / ((sigma ** 2 + 1) ** 0.5)
        return alpha, sigma * alpha

    def step(self, model_output, timestep, sample):
        i = self._step
        lower_final = i == len(self.timesteps) - 1 and len(self.timesteps) < 15
        alpha_t, sigma_t = self._alpha_sigma(self.sigmas[i])
        x0 = (sample - sigma_t * model_output) / alpha_t
        for k in range(self.solver_order - 1):
            self.model_outputs[k] = self.model_outputs[k + 1]
        self.model_outputs[-1] = x0
        a_t, s_t = self._alpha_sigma(self.sigmas[i + 1])
        a_s0, s_s0 = self._alpha_sigma(self.sigmas[i])
        lam_t, lam_s0 = torch.log(a_t) - torch.log(s_t), torch.log(a_s0) - torch.log(s_s0)
        h = lam_t - lam_s0
        if self.solver_order == 1 or self.lower_order_nums < 1 or lower_final:
            prev = (s_t / s_s0) * sample - (a_t * (torch.exp(-h) - 1.0)) * x0
        else:
            a_s1, s_s1 = self._alpha_sigma(self.sigmas[i - 1])
            lam_s1 = torch.log(a_s1) - torch.log(s_s1)
            m0, m1 = self.model_outputs[-1], self.model_outputs[-2]
            r0 = (lam_s0 - lam_s1) / h
            d0, d1 = m0, (1.0 / r0) * (m0 - m1)
            prev = (s_t / s_s0) * sample - (a_t * (torch.exp(-h) - 1.0)) * d0 - 0.5 * (a_t * (torch.exp(-h) - 1.0)) * d1
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self._step += 1
        return prev, x0


class PNDM(_Base):
    """skip_prk_steps=False, set_alpha_to_one=False, 'leading' spacing, steps_offset 0 (diffusers' defaults)."""

    def __init__(self, beta_start=0.00085, beta_end=0.012, beta_schedule="linear"):
        super().__init__(beta_start, beta_end, beta_schedule)
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.pndm_order = 4

    def set_timesteps(self, num_inference_steps: int):
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        self._timesteps = (np.arange(0, num_inference_steps) * ratio).round()
        prk = np.array(self._timesteps[-self.pndm_order:]).repeat(2) + np.tile(
            np.array([0, self.num_train_timesteps // num_inference_steps // 2]), self.pndm_order)
        self.prk_timesteps = (prk[:-1].repeat(2)[1:-1])[::-1].copy()
        self.plms_timesteps = self._timesteps[:-3][::-1].copy()
        self.timesteps = torch.from_numpy(np.concatenate([self.prk_timesteps, self.plms_timesteps]).astype(np.int64))
        self.ets, self.counter, self.cur_model_output, self.cur_sample = [], 0, 0, None

    def _prev(self, sample, timestep, prev_timestep, model_output):
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        b_t, b_prev = 1 - a_t, 1 - a_prev
        sample_coeff = (a_prev / a_t) ** 0.5
        denom = a_t * b_prev ** 0.5 + (a_t * b_t * a_prev) ** 0.5
        return sample_coeff * sample - (a_prev - a_t) * model_output / denom

    def step(self, model_output, timestep, sample):
        timestep = int(timestep)
        ratio = self.num_train_timesteps // self.num_inference_steps
        if self.counter < len(self.prk_timesteps):
            diff = 0 if self.counter % 2 else ratio // 2
            prev_timestep = timestep - diff
            timestep = int(self.prk_timesteps[self.counter // 4 * 4])
            if self.counter % 4 == 0:
                self.cur_model_output = self.cur_model_output + 1 / 6 * model_output
                self.ets.append(model_output)
                self.cur_sample = sample
            elif (self.counter - 1) % 4 == 0:
                self.cur_model_output = self.cur_model_output + 1 / 3 * model_output
            elif (self.counter - 2) % 4 == 0:
                self.cur_model_output = self.cur_model_output + 1 / 3 * model_output
            elif (self.counter - 3) % 4 == 0:
                model_output = self.cur_model_output + 1 / 6 * model_output
                self.cur_model_output = 0
            prev = self._prev(self.cur_sample, timestep, prev_timestep, model_output)
        else:
            prev_timestep = timestep - ratio
            self.ets = self.ets[-3:]
            self.ets.append(model_output)
            e = self.ets
            if len(e) == 1:
                m = e[-1]
            elif len(e) == 2:
                m = (3 * e[-1] - e[-2]) / 2
            elif len(e) == 3:
                m = (23 * e[-1] - 16 * e[-2] + 5 * e[-3]) / 12
            else:
                m = (1 / 24) * (55 * e[-1] - 59 * e[-2] + 37 * e[-3] - 9 * e[-4])
            prev = self._prev(sample, timestep, prev_timestep, m)
        self.counter += 1
        return prev, None
