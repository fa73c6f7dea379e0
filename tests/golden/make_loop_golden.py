"""Loop-level fixtures captured from the REFERENCE's own `ControlAnimationPipeline.__call__`
(animatediff/pipelines/controlanimation_pipeline.py:625-872, incl. prepare_latents :549-613, get_timesteps :615-622,
the ControlNet input selection :811-813) and its `MultiControlNetResidualsPipeline.__call__`
(modules/controlresiduals_pipeline.py:278-316).  Container only (SURVEY 8c last row).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_loop_golden.py

What runs as REFERENCE code: the whole `__call__` (prompt batching, timestep selection, latent preparation with its RNG
consumption, the loop, CFG combine), the in-file LCMScheduler (scenario custom_lcm), the UNet3D, the ControlNet-stack
wrapper (rearranges, prompt tiling quirk, fp16 casts).  What is stood in (none of it exists in /root/reference):
  * diffusers DDIM / LCM schedulers -> oracle/schedulers.py behind the diffusers call surface (so those two scenarios pin
    the LOOP, not the scheduler arithmetic);
  * diffusers MultiControlNetModel -> per-net oracle.controlnet.controlnet_forward (pinned by controlnet_refblocks.npz) + sum;
  * VAE / VaeImageProcessor -> tests/golden/loop_stubs.py (deterministic, replayed identically by the tests);
  * DiffusionPipeline plumbing (`_execution_device`, `progress_bar`).
"""
from __future__ import annotations

import contextlib
import functools
import os
import sys
from types import SimpleNamespace

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _refstub  # noqa: E402

_refstub.install()

import numpy as np  # noqa: E402
import torch  # noqa: E402

from loop_stubs import SCENARIOS, SMALL, PX, StubImageProcessor, StubVAE, scenario_inputs  # noqa: E402
from make_golden import ref_unet  # noqa: E402  (reference UNet3D builder)
from oracle import schedulers as OS  # noqa: E402
from oracle.controlnet import ControlNetConfig, controlnet_forward, init_controlnet_weights  # noqa: E402
from oracle.unet3d import UNet3DConfig, init_unet3d_weights  # noqa: E402

from animatediff.pipelines.controlanimation_pipeline import ControlAnimationPipeline, LCMScheduler  # noqa: E402  (reference)
from modules.controlresiduals_pipeline import MultiControlNetResidualsPipeline  # noqa: E402  (reference)


class DiffusersSchedulerFace:
    """The diffusers scheduler call surface the reference loop uses, over an oracle scheduler."""
    order = 1

    def __init__(self, inner):
        self.inner = inner
        self.config = SimpleNamespace()

    @property
    def timesteps(self):
        return self.inner.timesteps

    @property
    def init_noise_sigma(self):
        s = getattr(self.inner, "init_noise_sigma", 1.0)
        return s() if callable(s) else s

    def set_timesteps(self, num_inference_steps=None, device=None):
        self.inner.set_timesteps(num_inference_steps)

    def scale_model_input(self, sample, t=None):
        return self.inner.scale_model_input(sample, t)

    def add_noise(self, original, noise, timesteps):
        return self.inner.add_noise(original, noise, timesteps)


class DDIMFace(DiffusersSchedulerFace):
    def step(self, model_output, timestep, sample, eta=0.0, generator=None):
        prev, _ = self.inner.step(model_output, timestep, sample)
        return SimpleNamespace(prev_sample=prev)


class LCMFace(DiffusersSchedulerFace):
    def step(self, model_output, timestep, sample, generator=None):
        prev, den = self.inner.step(model_output, timestep, sample, generator=generator)
        return SimpleNamespace(prev_sample=prev, denoised=den)


class RefResiduals(MultiControlNetResidualsPipeline):
    """The reference wrapper with its constructor (HF downloads, annotators) and image preparation (PIL, cuda) replaced;
    `__call__` (:278-316) is the reference's."""

    def __init__(self, nets, cfg, cond_scale, use_lcm):
        self.nets, self.cfg, self.cond_scale, self.use_lcm = nets, cfg, cond_scale, use_lcm
        self.controlnets = nets
        self.controlnet_names = [f"synthetic-{i}" for i in range(len(nets))]
        self.calls = []

        def multi(sample, t, encoder_hidden_states=None, controlnet_cond=None, conditioning_scale=None, guess_mode=False, return_dict=False):
            self.calls.append(dict(sample_shape=tuple(sample.shape), dtype=str(sample.dtype), ehs_shape=tuple(encoder_hidden_states.shape),
                                   ehs_first_rows=encoder_hidden_states[:4, 0, :4].float().clone(), guess_mode=bool(guess_mode)))
            down_sum = mid_sum = None
            for sd, img, sc in zip(self.nets, controlnet_cond, conditioning_scale):  # MultiControlNetModel.forward: sum over nets
                d, m = controlnet_forward(sd, self.cfg, sample.float(), t, encoder_hidden_states.float(), img, sc, guess_mode)
                down_sum, mid_sum = (d, m) if down_sum is None else ([a + b for a, b in zip(down_sum, d)], mid_sum + m)
            return down_sum, mid_sum
        self.controlnet = multi

    def prep_control_images(self, images, control_image_processor=None, epoch=0, output_dir="", save_outputs=False,
                            do_classifier_free_guidance=True, guess_mode=False):
        ctrl = torch.stack([im.float() for im in images])  # hints in [0,1] (do_normalize=False, :160-163)
        self.prep_args = dict(do_cfg=bool(do_classifier_free_guidance), guess_mode=bool(guess_mode))
        out = []
        for _ in self.nets:
            c = ctrl
            if do_classifier_free_guidance and not guess_mode and not self.use_lcm:  # reference :268-269
                c = torch.cat([c] * 2)
            out.append(c)
        self.prep_images = out


def build_pipe(sc):
    pipe = object.__new__(ControlAnimationPipeline)  # (DiffusionPipeline.__init__/register_modules are not in /root/reference)
    if sc["unet"] == "lcm":
        cfg = UNet3DConfig.v2(block_out_channels=SMALL, time_cond_proj_dim=256)
        wseed = 3
    else:
        cfg = UNet3DConfig.v2(block_out_channels=SMALL)
        wseed = 1
    w = init_unet3d_weights(cfg, seed=wseed)
    unet = ref_unet(cfg, "v2")
    unet.load_state_dict(w, strict=False)
    unet.in_channels = unet.config.in_channels  # (diffusers' ModelMixin.__getattr__ forwards config entries; the shim does not)
    pipe.unet = unet
    pipe.vae = StubVAE()
    pipe.text_encoder = pipe.tokenizer = None
    pipe.ip_adapter = None
    pipe.vae_scale_factor = 8
    pipe.image_processor = StubImageProcessor()
    pipe.control_image_processor = None
    kw = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="linear")
    if sc["scheduler"] is None:
        pipe.scheduler = LCMScheduler(beta_start=0.00085, beta_end=0.0120, beta_schedule="scaled_linear", prediction_type="epsilon")  # :95-101
    elif sc["scheduler"] == "DDIMScheduler":
        pipe.scheduler = DDIMFace(OS.DDIM(**kw))
    else:
        pipe.scheduler = LCMFace(OS.DiffusersLCM(**kw))
    return pipe, wseed


type(ControlAnimationPipeline)  # noqa
ControlAnimationPipeline._execution_device = property(lambda self: torch.device("cpu"))
ControlAnimationPipeline.progress_bar = lambda self, total=None: contextlib.nullcontext(SimpleNamespace(update=lambda *a: None))
ControlAnimationPipeline.decode_latents = lambda self, latents: latents.detach().float().numpy()  # (the VAE decode is outside the loop)


@torch.no_grad()
def run(name):
    sc = SCENARIOS[name]
    frames, last, pos, neg = scenario_inputs(name)
    pipe, wseed = build_pipe(sc)
    cn = None
    cn_cfg = ControlNetConfig(block_out_channels=SMALL)
    if sc["nets"]:
        nets = [init_controlnet_weights(cn_cfg, seed=60 + i) for i in range(sc["nets"])]
        cn = RefResiduals(nets, cn_cfg, sc["cond_scale"], sc["use_lcm"])
    rec = dict(eps=[], latents=[], unet_in=[])
    fwd = pipe.unet.forward

    def spy(sample, timestep, *a, **k):
        out = fwd(sample, timestep, *a, **k)
        rec["unet_in"].append(sample.detach().clone())
        rec["eps"].append((out[0] if isinstance(out, tuple) else out.sample).detach().clone())
        return out
    pipe.unet.forward = spy
    prep0 = pipe.prepare_latents

    def prep_spy(*a, **k):
        lat = prep0(*a, **k)
        rec["init_latents"] = lat.detach().clone()
        return lat
    pipe.prepare_latents = prep_spy
    step0 = pipe.scheduler.step

    @functools.wraps(step0)  # (prepare_extra_step_kwargs inspects the signature for `eta` / `generator`)
    def step_spy(*a, **k):
        o = step0(*a, **k)
        rec["latents"].append((o[0] if isinstance(o, tuple) else o.prev_sample).detach().clone())
        return o
    pipe.scheduler.step = step_spy

    torch.manual_seed(sc["seed"])  # modules/controlanimate_pipeline.py:129-130
    gen = torch.Generator().manual_seed(sc["seed"])
    out = pipe(video_length=sc["frames"], input_frames=frames, prompt=None, height=PX, width=PX, num_inference_steps=sc["steps"],
               strength=sc["strength"], guidance_scale=sc["guidance"], generator=gen, overlaps=sc["overlaps"],
               multicontrolnetresiduals_pipeline=cn, prompt_embeds=pos, negative_prompt_embeds=neg,
               last_output_frames=last if last else None, use_lcm=sc["use_lcm"], guess_mode=sc["guess_mode"],
               use_img2img=sc["use_img2img"], save_outputs=False)
    arrs = {"timesteps": np.asarray([int(t) for t in (pipe.scheduler.timesteps if sc["strength"] >= 1 or sc["use_lcm"] else
                                                         pipe.get_timesteps(sc["steps"], sc["strength"], "cpu")[0])]),
            "init_latents": rec["init_latents"].numpy(), "final": out.videos.numpy()}
    for i, (e, l, u) in enumerate(zip(rec["eps"], rec["latents"], rec["unet_in"])):
        arrs[f"eps{i}"], arrs[f"latents{i}"] = e.numpy(), l.numpy()
        arrs[f"unet_in_shape{i}"] = np.asarray(u.shape)
    arrs["n_steps"] = len(rec["eps"])
    arrs["unet_weight_seed"] = wseed
    if cn is not None:
        arrs["cn_sample_shape"] = np.asarray(cn.calls[0]["sample_shape"])
        arrs["cn_ehs_shape"] = np.asarray(cn.calls[0]["ehs_shape"])
        arrs["cn_ehs_first_rows"] = cn.calls[0]["ehs_first_rows"].numpy()   # which prompt row each of the first 4 images saw
        arrs["cn_dtype_is_half"] = np.asarray(cn.calls[0]["dtype"] == "torch.float16")
        arrs["cn_prep_shape"] = np.asarray(cn.prep_images[0].shape)
    return arrs


if __name__ == "__main__":
    if not os.path.isdir("/root/reference"):
        raise SystemExit("needs /root/reference (container only)")
    torch.set_num_threads(8)
    allarr = {}
    for name in SCENARIOS:
        a = run(name)
        print(name, "steps", int(a["n_steps"]), "timesteps", a["timesteps"].tolist())
        for k, v in a.items():
            allarr[f"{name}/{k}"] = v
    np.savez_compressed(os.path.join(HERE, "loop_reference.npz"), **allarr)
    print("wrote loop_reference.npz", len(allarr), "arrays,", sum(np.asarray(v).nbytes for v in allarr.values()) // 1024, "KB raw")
