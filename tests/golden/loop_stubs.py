"""Deterministic stand-ins shared by the loop-fixture generator (make_loop_golden.py, run against the REFERENCE's own
ControlAnimationPipeline.__call__) and by the tests that replay the same scenarios through the oracle loop and the HIP
pipeline.  They replace the parts of the pipeline that are outside the denoising loop (VAE, image pre-processing):
plain data plumbing, no reference code."""
from __future__ import annotations

from types import SimpleNamespace

import torch
import torch.nn.functional as F


class _Dist:
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def sample(self, generator=None):
        # diffusers DiagonalGaussianDistribution.sample: randn_tensor(mean.shape, generator) -- CPU generator, CPU draw
        noise = torch.randn(self.mean.shape, generator=generator, dtype=torch.float32)
        return self.mean + self.std * noise.to(self.mean.device)


class StubVAE:
    """encode: 8x8 average pool + a fixed 3->4 channel map, std 0.05 (so the per-frame RNG draws of
    prepare_latents :577/:589 matter); decode: nearest x8 of 3 of the 4 channels.  scaling_factor 0.18215."""
    M = torch.tensor([[0.9, -0.3, 0.2], [-0.4, 0.8, 0.1], [0.3, 0.2, -0.7], [0.5, 0.5, 0.5]])

    def __init__(self):
        self.config = SimpleNamespace(scaling_factor=0.18215)
        self.dtype = torch.float32

    def encode(self, image):
        pooled = F.avg_pool2d(image.float(), 8)
        mean = torch.einsum("oc,bchw->bohw", self.M.to(image.device), pooled) * 2.0
        return SimpleNamespace(latent_dist=_Dist(mean, 0.05))

    def decode(self, latents):
        return SimpleNamespace(sample=F.interpolate(latents[:, :3].float(), scale_factor=8, mode="nearest"))


class StubImageProcessor:
    """VaeImageProcessor.preprocess for a [3,H,W] tensor in [0,1]: -> [1,3,H,W] in [-1,1]."""

    @staticmethod
    def preprocess(image):
        return (image.float() * 2.0 - 1.0)[None]


SCENARIOS = {
    # name: kwargs of __call__ + model variant
    "custom_lcm": dict(unet="lcm", scheduler=None, use_lcm=True, strength=0.5, steps=4, guidance=7.5, guess_mode=False,
                       nets=1, cond_scale=[0.8], frames=8, overlaps=0, last=0, use_img2img=True, seed=11),
    "ddim_cfg": dict(unet="v2", scheduler="DDIMScheduler", use_lcm=False, strength=1.0, steps=4, guidance=7.5, guess_mode=False,
                     nets=1, cond_scale=[1.0], frames=8, overlaps=0, last=0, use_img2img=True, seed=12),
    "lcm_lora_guess_overlap": dict(unet="v2", scheduler="LCMScheduler", use_lcm=False, strength=0.6, steps=5, guidance=1.5,
                                   guess_mode=True, nets=2, cond_scale=[1.0, 0.5], frames=8, overlaps=2, last=2, use_img2img=True, seed=13),
    "overlap_no_img2img": dict(unet="v2", scheduler="DDIMScheduler", use_lcm=False, strength=0.5, steps=4, guidance=7.5, guess_mode=False,
                               nets=0, cond_scale=[], frames=8, overlaps=3, last=3, use_img2img=False, seed=14),
}
SMALL = (64, 128, 256, 256)
PX = 64  # frame size in pixels (latent 8x8)


def scenario_inputs(name: str):
    """Seeded inputs of a scenario: frames / last output frames ([3,H,W] in [0,1]), prompt embeddings."""
    sc = SCENARIOS[name]
    g = torch.Generator().manual_seed(1000 + sc["seed"])
    frames = [torch.rand(3, PX, PX, generator=g) for _ in range(sc["frames"])]
    last = [torch.rand(3, PX, PX, generator=g) for _ in range(sc["last"])]
    pos = torch.randn(1, 77, 768, generator=g) * 0.5
    neg = torch.randn(1, 77, 768, generator=g) * 0.5
    return frames, last, pos, neg
