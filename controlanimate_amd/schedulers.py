"""Samplers of the denoising loop as host-side coefficient tables for ONE fused device kernel.

Every sampler the reference can select (modules/controlanimate_pipeline.py:52-67) is an affine
update in (sample, eps, noise) once the timestep is fixed; the loop therefore runs
`ca_cfg_scheduler_step` (CFG combine + update, include/controlanimate_hip.h) with 7 coefficients:

    x0   = (x - c0*eps) * c1          (optionally clamped)
    den  = c2*x0 + c3*x
    prev = c4*den + c5*eps + c6*noise

Implemented: the reference's in-tree native-LCM scheduler (`LCMScheduler` here, as in
animatediff/pipelines/controlanimation_pipeline.py:977,1375-1633) and all seven diffusers 0.23.0 classes of the
reference's table (modules/controlanimate_pipeline.py:52-61) with the constructor defaults it relies on
(`schedulers[config.scheduler](**noise_scheduler_kwargs)`): DDIMScheduler, LCMScheduler (exported as
`DiffusersLCMScheduler`), EulerDiscreteScheduler and EulerAncestralDiscreteScheduler go through the one fused kernel;
DPMSolverMultistepScheduler, LMSDiscreteScheduler and PNDMScheduler carry history (`multistep = True`): the loop asks
the kernel for the CFG-combined eps only and `step_device` forms the update as ONE `ca_lincomb` launch over the
current sample, the current / stored model outputs and noise, again with host-computed coefficients.
All scalar math is done in float64 on the host.  (diffusers is third-party and absent: parity of these restatements
is unpinned; oracle/schedulers.py holds an independent, step-function-shaped restatement they are tested against.)
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


class EulerAncestralDiscreteScheduler(EulerDiscreteScheduler):
    """diffusers 0.23.0 EulerAncestralDiscreteScheduler (defaults: 'linspace' spacing, epsilon prediction): the Euler step
    to sigma_down plus fresh noise of size sigma_up -- still affine in (sample, eps, noise): one fused launch."""
    needs_noise = True

    def coefficients(self, index: int):
        s_from, s_to = float(self.sigmas[index]), float(self.sigmas[index + 1])
        s_up = math.sqrt(s_to ** 2 * (s_from ** 2 - s_to ** 2) / s_from ** 2)
        s_down = math.sqrt(s_to ** 2 - s_up ** 2)
        # pred_x0 = x - sigma*eps; derivative = eps; prev = x + eps * (sigma_down - sigma) + noise * sigma_up
        return [0.0, 1.0, 0.0, 1.0, 1.0, s_down - s_from, s_up], 0.0


class _MultistepBase(_SchedulerBase):
    """Samplers with history.  `step_device(index, e, sample, noise, lincomb)` -> prev sample; `e` is the CFG-combined
    eps [1,c,f,h,w] fp32 on the device, `lincomb(terms)` = kernels.lincomb (sum_k coef_k * tensor_k in one launch)."""
    multistep = True

    def reset(self):
        raise NotImplementedError

    def coefficients(self, index: int):
        raise RuntimeError(f"{type(self).__name__} keeps history: the loop calls step_device(), not the fused 7-coefficient step")


class LMSDiscreteScheduler(_MultistepBase):
    """diffusers 0.23.0 LMSDiscreteScheduler (order 4, 'linspace' spacing, no Karras sigmas): prev = x + sum_j c_j * eps_{i-j}
    with c_j the integral of the j-th Lagrange basis polynomial over [sigma_i, sigma_{i+1}]."""

    def __init__(self, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", **kw):
        super().__init__(beta_start=beta_start, beta_end=beta_end, beta_schedule=beta_schedule, **kw)
        sig = np.sqrt((1 - self._ac) / self._ac)
        self.sigmas = np.concatenate([sig[::-1], [0.0]])
        self.init_noise_sigma = float(self.sigmas.max())
        self.derivatives: List[torch.Tensor] = []

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ts = np.linspace(0, self.num_train_timesteps - 1, num_inference_steps, dtype=float)[::-1].copy()
        sig = np.sqrt((1 - self._ac) / self._ac)
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        self.sigmas = np.concatenate([sig, [0.0]]).astype(np.float32).astype(np.float64)
        self.timesteps = torch.from_numpy(ts)
        self.init_noise_sigma = float(self.sigmas.max())
        self.reset()

    def reset(self):
        self.derivatives = []

    def input_scale(self, index: int) -> float:
        return 1.0 / math.sqrt(self.sigmas[index] ** 2 + 1.0)

    def lms_coefficient(self, order: int, t: int, current_order: int) -> float:
        from scipy import integrate

        def basis(tau):
            prod = 1.0
            for k in range(order):
                if k != current_order:
                    prod *= (tau - self.sigmas[t - k]) / (self.sigmas[t - current_order] - self.sigmas[t - k])
            return prod
        return integrate.quad(basis, self.sigmas[t], self.sigmas[t + 1], epsrel=1e-4)[0]

    def step_device(self, index, e, sample, noise, lincomb, order: int = 4):
        self.derivatives.append(e)          # derivative = (x - pred_x0) / sigma = eps
        if len(self.derivatives) > order:
            self.derivatives.pop(0)
        order = min(index + 1, order)
        coeffs = [self.lms_coefficient(order, index, k) for k in range(order)]
        return lincomb([(sample, 1.0)] + [(d, c) for c, d in zip(coeffs, reversed(self.derivatives))])


class DPMSolverMultistepScheduler(_MultistepBase):
    """diffusers 0.23.0 DPMSolverMultistepScheduler with its defaults: dpmsolver++, order 2, midpoint, lower_order_final,
    'linspace' spacing, sigma-space formulation (the last sigma is sigma(t=0), not 0)."""

    def __init__(self, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", solver_order=2, lower_order_final=True, **kw):
        super().__init__(beta_start=beta_start, beta_end=beta_end, beta_schedule=beta_schedule, **kw)
        self.solver_order, self.lower_order_final = solver_order, lower_order_final
        if solver_order not in (1, 2):
            raise NotImplementedError("solver_order 3 is not used by the reference (diffusers default: 2)")
        self.model_outputs: List[Optional[torch.Tensor]] = [None] * solver_order
        self.lower_order_nums = 0
        self.sigmas = np.sqrt((1 - self._ac) / self._ac)

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        last = self.num_train_timesteps  # lambda_min_clipped = -inf: nothing clipped
        ts = np.linspace(0, last - 1, num_inference_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
        sig = np.sqrt((1 - self._ac) / self._ac)
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        self.sigmas = np.concatenate([sig, [math.sqrt((1 - self._ac[0]) / self._ac[0])]]).astype(np.float32).astype(np.float64)
        self.timesteps = torch.from_numpy(ts)
        self.reset()

    def reset(self):
        self.model_outputs = [None] * self.solver_order
        self.lower_order_nums = 0

    @staticmethod
    def _alpha_sigma(sigma: float) -> Tuple[float, float]:
        alpha = 1.0 / math.sqrt(sigma ** 2 + 1.0)
        return alpha, sigma * alpha

    def step_device(self, index, e, sample, noise, lincomb):
        n = len(self.timesteps)
        final_low = index == n - 1 and self.lower_order_final and n < 15
        a_i, s_i = self._alpha_sigma(float(self.sigmas[index]))
        m0 = lincomb([(sample, 1.0 / a_i), (e, -s_i / a_i)])      # x0 prediction (dpmsolver++, epsilon model)
        self.model_outputs = self.model_outputs[1:] + [m0]
        a_t, s_t = self._alpha_sigma(float(self.sigmas[index + 1]))
        lam = lambda a, sg: math.log(a) - math.log(sg)
        h = lam(a_t, s_t) - lam(a_i, s_i)
        em1 = math.exp(-h) - 1.0
        if self.solver_order == 1 or self.lower_order_nums < 1 or final_low:
            prev = lincomb([(sample, s_t / s_i), (m0, -a_t * em1)])
        else:
            a_p, s_p = self._alpha_sigma(float(self.sigmas[index - 1]))
            r0 = (lam(a_i, s_i) - lam(a_p, s_p)) / h
            m1 = self.model_outputs[-2]
            # D0 = m0, D1 = (m0 - m1) / r0;  x_t = s_t/s_i x - a_t em1 D0 - 0.5 a_t em1 D1   (midpoint)
            prev = lincomb([(sample, s_t / s_i), (m0, -a_t * em1 * (1.0 + 0.5 / r0)), (m1, 0.5 * a_t * em1 / r0)])
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        return prev


class PNDMScheduler(_MultistepBase):
    """diffusers 0.23.0 PNDMScheduler with ITS defaults (the reference passes only the beta schedule): skip_prk_steps =
    False, set_alpha_to_one = False, 'leading' spacing -- i.e. three Runge-Kutta warm-up steps of four model
    evaluations each, then the linear multistep; `timesteps` therefore has 12 + (n - 3) entries for n steps."""

    def __init__(self, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", skip_prk_steps=False, set_alpha_to_one=False,
                 steps_offset=0, **kw):
        super().__init__(beta_start=beta_start, beta_end=beta_end, beta_schedule=beta_schedule, **kw)
        if skip_prk_steps:
            raise NotImplementedError("skip_prk_steps=True (not what the reference constructs)")
        self.final_alpha = 1.0 if set_alpha_to_one else float(self._ac[0])
        self.steps_offset = steps_offset
        self.pndm_order = 4
        self.reset()

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        base = (np.arange(0, num_inference_steps) * ratio).round() + self.steps_offset
        prk = np.array(base[-self.pndm_order:]).repeat(2) + np.tile(np.array([0, self.num_train_timesteps // num_inference_steps // 2]), self.pndm_order)
        self.prk_timesteps = (prk[:-1].repeat(2)[1:-1])[::-1].copy()
        self.plms_timesteps = base[:-3][::-1].copy()
        self.timesteps = torch.from_numpy(np.concatenate([self.prk_timesteps, self.plms_timesteps]).astype(np.int64))
        self.reset()

    def reset(self):
        self.ets: List[torch.Tensor] = []
        self.counter = 0
        self.cur_model_output = None   # running Runge-Kutta sum as a list of (tensor, coef)
        self.cur_sample = None

    def _prev_terms(self, sample, t: int, prev_t: int, model_terms):
        """_get_prev_sample (formula 9 of the PNDM paper) as lincomb terms; model_terms: [(tensor, coef)] of the model output."""
        a_t = self._ac[t]
        a_prev = self._ac[prev_t] if prev_t >= 0 else self.final_alpha
        denom = a_t * math.sqrt(1 - a_prev) + math.sqrt(a_t * (1 - a_t) * a_prev)
        k = -(a_prev - a_t) / denom
        return [(sample, math.sqrt(a_prev / a_t))] + [(x, c * k) for x, c in model_terms]

    def step_device(self, index, e, sample, noise, lincomb):
        ratio = self.num_train_timesteps // self.num_inference_steps
        t = int(self.timesteps[index])
        if self.counter < len(self.prk_timesteps):
            diff = 0 if self.counter % 2 else ratio // 2
            prev_t = t - diff
            t_eff = int(self.prk_timesteps[self.counter // 4 * 4])
            phase = self.counter % 4
            if phase == 0:
                self.cur_model_output = [(e, 1 / 6)]
                self.ets.append(e)
                self.cur_sample = sample
                model = [(e, 1.0)]
            elif phase in (1, 2):
                self.cur_model_output = self.cur_model_output + [(e, 1 / 3)]
                model = [(e, 1.0)]
            else:
                model = self.cur_model_output + [(e, 1 / 6)]
                self.cur_model_output = None
            prev = lincomb(self._prev_terms(self.cur_sample, t_eff, prev_t, model))
        else:
            prev_t = t - ratio
            self.ets = self.ets[-3:] + [e]
            w = {1: [1.0], 2: [3 / 2, -1 / 2], 3: [23 / 12, -16 / 12, 5 / 12], 4: [55 / 24, -59 / 24, 37 / 24, -9 / 24]}[len(self.ets)]
            prev = lincomb(self._prev_terms(sample, t, prev_t, [(x, c) for x, c in zip(reversed(self.ets), w)]))
        self.counter += 1
        return prev


SCHEDULERS = {
    "DDIMScheduler": DDIMScheduler,
    "LCMScheduler": DiffusersLCMScheduler,  # config name -> diffusers' class
    "EulerDiscreteScheduler": EulerDiscreteScheduler,
    "EulerAncestralDiscreteScheduler": EulerAncestralDiscreteScheduler,
    "DPMSolverMultistepScheduler": DPMSolverMultistepScheduler,
    "LMSDiscreteScheduler": LMSDiscreteScheduler,
    "PNDMScheduler": PNDMScheduler,
}


def get_scheduler(name: str, **kwargs):
    if name not in SCHEDULERS:
        raise NotImplementedError(f"scheduler {name!r} is not one of the reference's table: {sorted(SCHEDULERS)}")
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
