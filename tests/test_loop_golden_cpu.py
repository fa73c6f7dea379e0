"""CPU: loop-level fixtures captured from the REFERENCE's own ControlAnimationPipeline.__call__
(tests/golden/make_loop_golden.py -> loop_reference.npz).

  * the product's `prepare_latents` (host logic: RNG consumption order, img2img / overlap latent init
    reference :549-613) against the reference's initial latents -- bit-for-bit inputs, fp32 arithmetic;
  * the oracle loop (oracle/denoise_loop.py) against the reference's per-step eps and latents: pins the CFG batching,
    the ControlNet input selection (:811-813), the prompt-tiling quirk, get_timesteps, the decode-`denoised` rule.
"""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from loop_stubs import PX, SCENARIOS, SMALL, StubVAE, scenario_inputs  # noqa: E402

FX = np.load(os.path.join(HERE, "golden", "loop_reference.npz"))


def fx(name, key):
    return torch.from_numpy(np.asarray(FX[f"{name}/{key}"]))


def product_scheduler(sc):
    from controlanimate_amd.configs import NOISE_SCHEDULER_KWARGS
    from controlanimate_amd.schedulers import get_scheduler
    return None if sc["scheduler"] is None else get_scheduler(sc["scheduler"], **NOISE_SCHEDULER_KWARGS)


def product_prepare(name, gen):
    """The product pipeline's timestep selection + prepare_latents on CPU (no kernels involved)."""
    from controlanimate_amd.controlanimation_pipeline import ControlAnimationPipeline
    sc = SCENARIOS[name]
    frames, last, _, _ = scenario_inputs(name)
    pipe = ControlAnimationPipeline(vae=StubVAE(), text_encoder=None, tokenizer=None, unet=None, scheduler=product_scheduler(sc))
    pipe.device = torch.device("cpu")
    sched = pipe.scheduler
    if sc["use_lcm"]:
        sched.set_timesteps(sc["strength"], sc["steps"], 50)
        timesteps = sched.timesteps
    else:
        sched.set_timesteps(sc["steps"])
        timesteps = sched.timesteps if sc["strength"] >= 1 else pipe.get_timesteps(sc["steps"], sc["strength"])[0]
    lat = pipe.prepare_latents(frames, 1, 4, sc["frames"], PX, PX, torch.float32, "cpu", gen, timesteps[:1], sc["overlaps"], sc["strength"],
                               None, last if last else None, sc["use_lcm"], sc["use_img2img"])
    return timesteps, lat


@pytest.mark.parametrize("name", list(SCENARIOS))
def test_prepare_latents_and_timesteps_match_the_reference(name):
    gen = torch.Generator().manual_seed(SCENARIOS[name]["seed"])
    timesteps, lat = product_prepare(name, gen)
    assert [int(t) for t in timesteps] == fx(name, "timesteps").tolist()
    ref = fx(name, "init_latents")
    assert lat.shape == ref.shape
    assert torch.allclose(lat, ref, atol=2e-6, rtol=1e-6), float((lat - ref).abs().max())
    if SCENARIOS[name]["overlaps"]:
        # the frames that continue the previous window really start from ITS latents (not from noise)
        noise_only = torch.randn(ref.shape, generator=torch.Generator().manual_seed(SCENARIOS[name]["seed"]))
        assert not torch.allclose(ref[:, :, 0], noise_only[:, :, 0], atol=1e-3)


@pytest.mark.parametrize("name", ["custom_lcm", "ddim_cfg", "lcm_lora_guess_overlap", "overlap_no_img2img"])
def test_oracle_loop_matches_the_reference_call(name):
    from oracle.controlnet import ControlNetConfig, init_controlnet_weights
    from oracle.denoise_loop import LoopInputs, denoise_loop
    from oracle.unet3d import UNet3DConfig, init_unet3d_weights
    sc = SCENARIOS[name]
    frames, last, pos, neg = scenario_inputs(name)
    ucfg = UNet3DConfig.v2(block_out_channels=SMALL, **({"time_cond_proj_dim": 256} if sc["unet"] == "lcm" else {}))
    uw = init_unet3d_weights(ucfg, seed=int(FX[f"{name}/unet_weight_seed"]))
    ccfg = ControlNetConfig(block_out_channels=SMALL)
    nets = [init_controlnet_weights(ccfg, seed=60 + i) for i in range(sc["nets"])]
    torch.manual_seed(sc["seed"])                      # global RNG: the in-tree LCMScheduler's step noise (:1601)
    gen = torch.Generator().manual_seed(sc["seed"])    # CPU generator: latents, VAE samples, diffusers-LCM step noise
    _, _ = product_prepare(name, gen)                  # (advances `gen` exactly as the reference's prepare_latents does)
    n = int(FX[f"{name}/n_steps"])
    step_noises = None
    if sc["scheduler"] == "LCMScheduler":
        step_noises = [torch.randn(fx(name, "init_latents").shape, generator=gen) for _ in range(n)]
    inp = LoopInputs(latents=fx(name, "init_latents"), prompt_embeds=pos, negative_prompt_embeds=neg, guidance_scale=sc["guidance"],
                     num_inference_steps=sc["steps"], scheduler=sc["scheduler"] or "custom_lcm", strength=sc["strength"], use_lcm=sc["use_lcm"],
                     guess_mode=sc["guess_mode"], control_images=[torch.stack(frames)] * sc["nets"] if sc["nets"] else None,
                     cond_scale=sc["cond_scale"], step_noises=step_noises)
    out = denoise_loop(uw, ucfg, inp, nets or None, ccfg)
    assert [int(t) for t in out["timesteps"]] == fx(name, "timesteps").tolist()
    assert len(out["eps"]) == n
    for i in range(n):
        raw = fx(name, f"eps{i}")       # the reference UNet's raw output (both CFG halves when b = 2)
        mine = out["eps_raw"][i]
        assert mine.shape == raw.shape == tuple(FX[f"{name}/unet_in_shape{i}"].tolist())
        rel = ((mine - raw).norm() / raw.norm()).item()
        # fp32 vs fp32.  Step 0 sees identical inputs; later steps inherit the (guidance-amplified) fp32 round-off of the
        # previous ones, and the fp16 cast of the ControlNet inputs (:295-297, mirrored) turns a 1e-7 input difference into a
        # 5e-4 one wherever it crosses a rounding boundary
        assert rel < (2e-4 if i == 0 else 2e-3), (i, rel)
        lrel = ((out["latents"][i] - fx(name, f"latents{i}")).norm() / fx(name, f"latents{i}").norm()).item()
        assert lrel < 2e-3, (i, lrel)
    fin = out["final"]
    assert ((fin - fx(name, "final")).norm() / fx(name, "final").norm()).item() < 2e-3


def test_controlnet_call_contract_recorded_from_the_reference():
    """What the reference hands its ControlNet stack (modules/controlresiduals_pipeline.py:278-316), as recorded by the
    fixture generator: batch layout, fp16 casts, CFG hint doubling, prompt tiling order."""
    f = SCENARIOS["ddim_cfg"]["frames"]
    # non-guess CFG: both halves go through the ControlNet; hints doubled; prompts tiled [neg,pos,neg,pos,...] (quirk C-1)
    assert FX["ddim_cfg/cn_sample_shape"].tolist() == [2 * f, 4, 8, 8]
    assert FX["ddim_cfg/cn_prep_shape"].tolist() == [2 * f, 3, PX, PX]
    _, _, pos, neg = scenario_inputs("ddim_cfg")
    rows = torch.from_numpy(FX["ddim_cfg/cn_ehs_first_rows"])
    want = torch.stack([neg[0, 0, :4], pos[0, 0, :4], neg[0, 0, :4], pos[0, 0, :4]]).half().float()
    assert torch.allclose(rows, want)
    assert bool(FX["ddim_cfg/cn_dtype_is_half"])
    # guess mode / native LCM: the un-doubled latents and the positive prompt only
    assert FX["lcm_lora_guess_overlap/cn_sample_shape"].tolist() == [f, 4, 8, 8]
    assert FX["lcm_lora_guess_overlap/cn_prep_shape"].tolist() == [f, 3, PX, PX]
    assert FX["custom_lcm/cn_sample_shape"].tolist() == [f, 4, 8, 8]
