"""ControlAnimationPipeline: prompt/IP embeds, timesteps, initial latents and the DENOISING LOOP.

Call surface of the reference's animatediff/pipelines/controlanimation_pipeline.py
(ctor :76-92, __call__ :626-663 -> AnimationPipelineOutput(videos)); loop semantics of :790-855:

  per step   ControlNet input selection (:811-813)  ->  MultiControlNetResidualsPipeline
             UNet3D eps (CFG batch b=2 unless native LCM)                       (:821-841)
             CFG combine + scheduler update, ONE fused kernel (:844-849 / :833)
  skipped    torch.cuda.empty_cache() every step (:794) -- allocator flush + sync, no output effect.

The VAE is controlanimate_amd/vae.py; the CLIP text encoder stays out of scope.  The
pipeline therefore takes `prompt_embeds` / `negative_prompt_embeds` (as the reference's facade
already does, modules/controlanimate_pipeline.py:133-146); if a `vae` object with the diffusers
AutoencoderKL interface is supplied it is used for encode/decode exactly where the reference does,
otherwise frames are exchanged as latents (`input_latents`, `last_output_latents`,
`output_type="latent"`).
RNG parity: initial latents come from the caller's CPU torch.Generator; the in-tree LCM
sampler draws its per-step noise from the GLOBAL CPU RNG (:1601), diffusers' LCM from the generator.
"""
from __future__ import annotations

import logging
import time
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from . import kernels as K
from .controlresiduals_pipeline import MultiControlNetResidualsPipeline, _image_to_chw01
from .schedulers import DiffusersLCMScheduler, LCMScheduler, get_w_embedding

logger = logging.getLogger(__name__)


@dataclass
class AnimationPipelineOutput:
    videos: Union[torch.Tensor, np.ndarray]


class ControlAnimationPipeline:
    def __init__(self, vae, text_encoder, tokenizer, unet, scheduler=None):
        self.overlap_controlnet = True  # ControlNet stack on a second HIP stream beside the UNet encoder
        # True (the default): the ControlNet + UNet part of a step is captured ONCE as a hipGraph -- in the first window, after
        # an eager step 0 that warms the prompt / hint caches and the allocator pools -- and replayed for every later step of
        # EVERY later window with the same signature (shapes, models, sampler family): static input buffers, device-side
        # timestep, the window's prompt embeddings / control frames copied into the tensors the captured kernels read and
        # the per-window caches refreshed in place.  Same kernels, bit-identical results (tests/test_graph_gpu.py), ~1 ms
        # instead of ~50 ms of host time per step: with 8 ranks on one host the loop stays GPU-bound.  A failed capture is
        # logged and the window runs eagerly (`graph_fallback_reason`).
        self.use_hip_graph = True
        # True: a WHOLE window -- every loop iteration: CFG duplicate, ControlNet stack, UNet3D, CFG combine + sampler update -- is
        # ONE captured hipGraph, replayed once per window: the reference's loop (:792-855) has no host dependence between its steps
        # either.  Each step's timestep, input scale and sampler coefficients are constants of its own kernel nodes; the sampler
        # noise of the window is drawn and uploaded before the replay.  Taken when nothing needs the host between steps: no
        # `callback`, no `record_eps`, a single-step sampler (the history-carrying ones keep the per-step graph) and a whole window
        # (`step_range` unset or full); otherwise the per-step graph above.  Bit-identical latents (tests/test_graph_gpu.py).
        # OFF by default, by measurement (round 6, ROCm 7.2, one box, config 2): hipGraphLaunch's host cost grows with the SQUARE of
        # the node count -- 3 ms for the 786 kernels of one step, 1.1 s for the 15 720 of a 20-step window -- so the one-replay window
        # is host-bound: 65.9 / 66.2 ms per step against 49.8 / 49.7 for twenty per-step replays (profiles/round6_ab_window_graph.json).
        self.window_graph = False
        self.window_graph_fallback_reason = None
        self.window_replays = 0  # whole-window replays since construction (tests, bench)
        self.graph_recapture_reason = None  # why the last call dropped a captured graph (None: it did not)
        # How far the host may run ahead of the device, in denoise steps.  An unpaced loop enqueues faster than the device
        # drains (a replay is ~3 us of host time per kernel against ~75 us of device time), fills the hardware queue and then
        # SPINS inside the runtime for the rest of every step: two cores per rank for nothing (round 4: 122 ms of CPU per
        # 59 ms step).  With a bound the loop waits until the step `steps_in_flight` steps back has finished before it
        # enqueues the next one; 2 keeps one whole step queued behind the running one, so the device never waits for the
        # host.  HOW it waits is `pace_wait`: "sleep" polls hipEventQuery between `time.sleep(pace_poll_s)` naps -- the
        # thread is off the CPU (measured on this pool: hipEventSynchronize spins even on a hipEventBlockingSync event, see
        # DESIGN 5); "event" is that blocking synchronize, kept for A/B runs.  0 = unpaced (the round-4 behaviour).  The end
        # of a window is awaited the same way.
        self.pace_wait = "sleep"
        self.pace_poll_s = 0.001
        self.pace_timeout_s = 120.0
        self.steps_in_flight = 2
        self.fuse_controlnet_adds = True  # (False: separate ca_add_bcast passes, as round 2 -- A/B runs and the tests that compare the two)
        self._graph_state = None
        self._noise_state = None
        if scheduler is None:  # native LCM (reference :95-101)
            scheduler = LCMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", prediction_type="epsilon")
        self.vae, self.text_encoder, self.tokenizer, self.unet, self.scheduler = vae, text_encoder, tokenizer, unet, scheduler
        vcfg = getattr(vae, "config", None)
        boc = vcfg.get("block_out_channels") if isinstance(vcfg, dict) else getattr(vcfg, "block_out_channels", None)
        self.vae_scale_factor = 2 ** (len(boc) - 1) if boc else 8  # reference :158 (SD1.5: 8)
        self.ip_adapter = None
        from .local_models import VaeImageProcessor  # reference :159-163
        self.image_processor = VaeImageProcessor(vae_scale_factor=self.vae_scale_factor, do_convert_rgb=True)
        self.control_image_processor = VaeImageProcessor(vae_scale_factor=self.vae_scale_factor, do_convert_rgb=True, do_normalize=False)
        self._pending_lora: List[dict] = []
        self.device = torch.device("cuda")
        self.last_step_times: List[float] = []
        self.single_host_thread = True  # __call__ runs with ONE torch intra-op thread (restored on return): see __call__
        self.record_eps = False      # True: keep every step's raw UNet eps ([rep,4,f,h,w] fp32, CPU) in `eps_history` (parity tests)
        self.eps_history: List[torch.Tensor] = []

    def to(self, device):
        self.device = torch.device(device)
        if self.unet is not None:
            self.unet.to(self.device)
        if hasattr(self.vae, "to"):
            self.vae.to(self.device)
        if isinstance(self.text_encoder, torch.nn.Module):
            self.text_encoder.to(self.device)
        return self

    @property
    def _execution_device(self):
        return self.device

    # ---- loader mixins the reference inherits from diffusers and calls on the pipeline -------------------------------
    def maybe_convert_prompt(self, prompt, tokenizer=None):
        """TextualInversionLoaderMixin.maybe_convert_prompt (used at modules/controlanimate_pipeline.py:120-121)."""
        from .local_models import maybe_convert_prompt
        return maybe_convert_prompt(prompt, tokenizer if tokenizer is not None else self.tokenizer)

    def load_textual_inversion(self, pretrained_model_name_or_path: str, token: Optional[str] = None, **_):
        """TextualInversionLoaderMixin.load_textual_inversion for one embedding file (:118): the (multi-vector)
        embedding becomes new tokenizer entries token, token_1, ... and new rows of the text encoder's token table."""
        from .local_models import read_textual_inversion
        if self.tokenizer is None or self.text_encoder is None:
            raise RuntimeError("load_textual_inversion needs a tokenizer and a text_encoder")
        tokens, emb = read_textual_inversion(pretrained_model_name_or_path, token)
        vocab = self.tokenizer.get_vocab()
        clash = [t for t in tokens if t in vocab]
        if clash:
            raise ValueError(f"token(s) {clash} already in the tokenizer vocabulary; pass a different `token`")
        self.tokenizer.add_tokens(tokens)
        table = self.text_encoder.text_model.embeddings.token_embedding
        if emb.shape[1] != table.weight.shape[1]:
            raise ValueError(f"embedding width {emb.shape[1]} != text encoder width {table.weight.shape[1]}")
        old = table.weight.data
        new_rows = len(self.tokenizer) - old.shape[0]
        grown = torch.cat([old, torch.zeros(max(new_rows, 0), old.shape[1], dtype=old.dtype, device=old.device)])
        for t, e in zip(tokens, emb):
            grown[self.tokenizer.convert_tokens_to_ids(t)] = e.to(grown)
        table.weight = torch.nn.Parameter(grown, requires_grad=False)
        table.num_embeddings = grown.shape[0]
        if hasattr(self.text_encoder, "config"):
            self.text_encoder.config.vocab_size = grown.shape[0]
        if getattr(self.text_encoder, "arena", None) is not None:
            self.text_encoder.prepare(self.device)  # repack
        return tokens

    def load_lora_weights(self, pretrained_model_name_or_path_or_dict, **_):
        """LoraLoaderMixin.load_lora_weights (animatediff/utils/util.py:155): the state dict is kept until fuse_lora."""
        from .weight_ingest import read_checkpoint
        sd = pretrained_model_name_or_path_or_dict
        self._pending_lora.append(dict(sd) if isinstance(sd, dict) else read_checkpoint(sd))

    def fuse_lora(self, lora_scale: float = 1.0, **_):
        """LoraLoaderMixin.fuse_lora (util.py:156): W += scale * alpha / rank * up @ down on the UNet and, when the
        file carries them and a fusable text encoder is attached, on the text encoder."""
        from .weight_ingest import fuse_lora, fuse_lora_text_encoder, lora_text_encoder_keys
        for sd in self._pending_lora:
            fuse_lora(self.unet, sd, lora_scale=float(lora_scale))
            if lora_text_encoder_keys(sd):
                if isinstance(self.text_encoder, torch.nn.Module):
                    fuse_lora_text_encoder(self.text_encoder, sd, lora_scale=float(lora_scale))
                else:
                    logger.warning("text-encoder LoRA tensors skipped: no fusable text_encoder attached")
        self._pending_lora = []
        for m in (self.unet, self.text_encoder):
            if m is not None and getattr(m, "arena", None) is not None:
                m.prepare(self.device)

    def enable_xformers_memory_efficient_attention(self):
        """No-op: the flash attention kernel is the only attention path (no xformers)."""

    def get_w_embedding(self, w, embedding_dim=512, dtype=torch.float32):
        return get_w_embedding(w, embedding_dim, dtype)

    def get_timesteps(self, num_inference_steps, strength, device=None):
        init_timestep = min(int(num_inference_steps * strength), num_inference_steps)
        t_start = max(num_inference_steps - init_timestep, 0)
        return self.scheduler.timesteps[t_start * self.scheduler.order:], num_inference_steps - t_start

    # ------------------------------------------------------------------------------------------
    def _encode_frames(self, frames, generator) -> List[torch.Tensor]:
        if self.vae is None:
            raise RuntimeError("frames given as images need a `vae` (VAE encode is a 'next' component); pass latents instead")
        out = []
        if hasattr(self.vae, "encode_moments") and len(frames) > 0:
            # HIP VAE: one batched encoder pass for all frames; the Gaussian sample is still drawn frame by
            # frame from the same generator, i.e. the reference's RNG consumption (:577, :589)
            from .vae import DiagonalGaussianDistribution
            imgs = torch.stack([_image_to_chw01(fr) * 2.0 - 1.0 for fr in frames]).to(self.device)
            mean, logvar = self.vae.encode_moments(imgs)
            for i in range(len(frames)):
                lat = DiagonalGaussianDistribution(mean[i:i + 1], logvar[i:i + 1]).sample(generator)
                out.append(self.vae.config.scaling_factor * lat.float())
            return out
        for fr in frames:
            img = (_image_to_chw01(fr) * 2.0 - 1.0)[None].to(self.device)
            lat = self.vae.encode(img).latent_dist.sample(generator)
            out.append(self.vae.config.scaling_factor * lat.float())
        return out

    def prepare_latents(self, input_frames, batch_size, num_channels_latents, video_length, height, width, dtype, device,
                        generator, latent_timestep, overlaps, strength, latents=None, last_output_frames=None, use_lcm=False,
                        use_img2img=False, input_latents=None, last_output_latents=None):
        """Reference :549-613. Returns fp32 [1,4,f,h/8,w/8] on `device`. `input_latents` [1,4,f,h,w] /
        `last_output_latents` [1,4,n,h,w] replace the per-frame VAE encodes when no VAE is attached."""
        shape = (batch_size, num_channels_latents, video_length, height // self.vae_scale_factor, width // self.vae_scale_factor)
        noise = torch.randn(shape, generator=generator, dtype=torch.float32)  # randn_tensor with a CPU generator
        latents = noise.clone()
        if overlaps > 0 or strength < 1 or use_lcm:
            frames_lat = None
            if input_latents is not None:
                frames_lat = [input_latents[:, :, i].float().cpu() for i in range(input_latents.shape[2])]
            elif input_frames is not None and self.vae is not None:
                frames_lat = [x.cpu() for x in self._encode_frames(input_frames, generator)]
            last_lat = None
            if last_output_latents is not None:
                last_lat = [last_output_latents[:, :, i].float().cpu() for i in range(last_output_latents.shape[2])]
            elif last_output_frames is not None and self.vae is not None:
                last_lat = [x.cpu() for x in self._encode_frames(last_output_frames, generator)]
            if use_lcm:
                if frames_lat is None:
                    raise RuntimeError("native LCM starts from the input frames' latents (reference :590-592)")
                latents = self.scheduler.add_noise(torch.stack(frames_lat, dim=2), latents, latent_timestep)
            elif last_lat is not None and strength < 1.0:
                for i in range(video_length):
                    if i < len(last_lat):
                        src = last_lat[i]
                    elif not use_img2img:
                        src = last_lat[-1]
                    else:
                        src = frames_lat[i]
                    latents[:, :, i] = self.scheduler.add_noise(src, latents[:, :, i], latent_timestep)
        if strength >= 1 and not use_lcm:
            latents = latents * self.scheduler.init_noise_sigma
        return latents.to(device)

    def decode_latents(self, latents):
        if self.vae is None:
            raise RuntimeError("no vae attached: request output_type='latent'")
        video_length = latents.shape[2]
        latents = 1 / self.vae.config.scaling_factor * latents
        if hasattr(self.vae, "encode_moments"):  # HIP VAE: all frames of the window in one batch (no cross-frame op)
            b, c, f, h, w = latents.shape
            flat = latents.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
            video = self.vae.decode(flat).sample
            video = video.view(b, f, *video.shape[1:]).permute(0, 2, 1, 3, 4)
        else:
            frames = [self.vae.decode(latents[:, :, i]).sample for i in range(video_length)]
            video = torch.stack(frames, dim=2)
        return (video / 2 + 0.5).clamp(0, 1).cpu().float().numpy()

    def release_graph(self) -> None:
        """Drops the captured hipGraph with its private memory pool and every object its signature keeps alive (models, weight
        arenas, per-window caches, control images).  Called by __call__ when a call runs without replay; call it yourself after
        swapping or releasing models.  The next graph-mode call captures again."""
        self._graph_state = None

    def _await(self, ev) -> None:
        """Waits for a recorded pacing event without burning a core (see `steps_in_flight`)."""
        if self.pace_wait == "event":
            ev.synchronize()
            return
        nap, t0 = float(self.pace_poll_s), time.monotonic()
        limit = float(self.pace_timeout_s) * max(1, int(self.steps_in_flight))  # (the awaited event sits behind that many steps)
        while not ev.query():
            time.sleep(nap)
            if time.monotonic() - t0 > limit:  # a device hang must not become a silent endless poll
                # the work recorded before the event is still queued: nothing of this pipeline may be reused as it is -- the pacing
                # ring's events (a later call would wait on them out of order), the captured graph and the noise buffers in flight
                self.__dict__.pop("_pace_ring", None)
                self._graph_state = None
                self._noise_state = None
                raise RuntimeError(f"the denoise step recorded {limit:.0f} s ago has not finished (device hang? serialised profiling and many "
                                   f"ranks on one GPU stretch a step: raise `pace_timeout_s`, now {self.pace_timeout_s:.0f} s per step in flight)")

    def _pace_event(self, i: int):
        """Events of the pacing ring (created once: an event per step would be 20 create / destroy pairs per window).  Plain events for
        the polling wait; hipEventBlockingSync ones only for pace_wait = "event"."""
        kind = self.pace_wait == "event"
        ring = self.__dict__.setdefault("_pace_ring", {}).setdefault(kind, [])
        n = max(1, int(self.steps_in_flight)) + 1
        while len(ring) < n:
            ring.append(torch.cuda.Event(blocking=kind))
        return ring[i % n]

    # ---- what a captured hipGraph of the step reads besides its own pool ---------------------------------------------
    @staticmethod
    def _model_cache_owners(gs, unet, nets, cn):
        """Strong references to the buffers a captured step reads but the pipeline does not allocate: the models' per-window
        caches (prompt copy + text / IP K/V dict, hint embeddings) and their weight arenas, as they are at capture time."""
        def prompt_slot(m, src):
            ent = m.cache_slot(src)
            c = {} if ent is None else ent["cache"]
            return c, c.get("ehs")
        def hint_slot(n, k_):
            ent = n.hint_slot(cn.prep_images[k_]) if cn is not None and cn.prep_images is not None else None
            return (None, None) if ent is None else (ent["emb"], ent["src"])
        return {"unet": (*prompt_slot(unet, gs["unet_prompt"]), unet.arena),
                "nets": [(*prompt_slot(n, gs["cn_prompt"]), *hint_slot(n, k_), n.arena) for k_, n in enumerate(nets)]}

    @staticmethod
    def _graph_lost_model_caches(gs, unet, nets, cn) -> Optional[str]:
        """None while every such buffer is still the object the graph was captured on AND still keyed on the pipeline's static prompt /
        control-image tensors (so that `refresh_window_caches` refreshes from the right source); else what changed.  The models keep one
        cache slot per prompt / control-image tensor: the graph owns ITS slots as long as they are there -- another pipeline's slots
        beside them do not matter."""
        own = gs.get("owned")
        if own is None:
            return "nothing recorded at capture time"
        c, ehs, arena = own["unet"]
        ent = unet.cache_slot(gs["unet_prompt"])
        if ent is None or ent["cache"] is not c or c.get("ehs") is not ehs:
            return "the UNet's prompt cache slot was replaced or evicted"
        if unet.arena is not arena:
            return "the UNet's weight arena was re-packed"
        if len(own["nets"]) != len(nets):
            return "another number of ControlNets"
        for k_, (n, (c, ehs, hint, hint_src, arena)) in enumerate(zip(nets, own["nets"])):
            ent = n.cache_slot(gs["cn_prompt"])
            if ent is None or ent["cache"] is not c or c.get("ehs") is not ehs:
                return f"ControlNet {k_}'s prompt cache slot was replaced or evicted"
            if n.arena is not arena:
                return f"ControlNet {k_}'s weight arena was re-packed"
            if cn is None or cn.prep_images is None or cn.prep_images[k_] is not hint_src:
                return f"ControlNet {k_}'s control-image tensor is another object"
            hent = n.hint_slot(hint_src)
            if hent is None or hent["emb"] is not hint:
                return f"ControlNet {k_}'s hint-embedding slot was replaced or evicted"
        return None

    @classmethod
    def _graph_owns_model_caches(cls, gs, unet, nets, cn) -> bool:
        return cls._graph_lost_model_caches(gs, unet, nets, cn) is None

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, video_length: Optional[int], input_frames: list = None, prompt=None, height: Optional[int] = None,
                 width: Optional[int] = None, num_inference_steps: int = 50, strength: float = 0.5, guidance_scale: float = 7.5,
                 negative_prompt=None, num_videos_per_prompt: Optional[int] = 1, eta: float = 0.0,
                 generator: Optional[torch.Generator] = None, latents: Optional[torch.Tensor] = None,
                 output_type: Optional[str] = "tensor", return_dict: bool = True,
                 callback: Optional[Callable[[int, int, torch.Tensor], None]] = None, callback_steps: Optional[int] = 1,
                 overlaps: int = 0, multicontrolnetresiduals_pipeline: Optional[MultiControlNetResidualsPipeline] = None,
                 multicontrolnetresiduals_overlap_pipeline=None, num_images_per_prompt: Optional[int] = 1, clip_skip=None,
                 cross_attention_kwargs: Optional[Dict[str, Any]] = None, prompt_embeds: Optional[torch.Tensor] = None,
                 negative_prompt_embeds: Optional[torch.Tensor] = None, epoch=0, output_dir="tmp/output", save_outputs=False,
                 last_output_frames=None, use_lcm=True, lcm_origin_steps: int = 50, guess_mode=False, ipa_scale=0.4,
                 use_img2img=True, input_latents: Optional[torch.Tensor] = None,
                 last_output_latents: Optional[torch.Tensor] = None, control_images=None, **kwargs):
        # The host side of the loop is a few scalar coefficient computations and the serial draw of the sampler's noise: nothing an
        # intra-op thread pool can help with -- but torch's OpenMP workers, once woken by any parallel region, spin for their
        # block time beside the calling thread (measured: ~26 ms of CPU per 53 ms step on a 16-core host, gone with OMP_NUM_THREADS=1).
        # The call therefore owns its host threads: one intra-op thread while it runs, the caller's setting restored afterwards.
        host_threads = torch.get_num_threads()
        if self.single_host_thread and host_threads != 1:
            torch.set_num_threads(1)
        try:
            return self._denoise(video_length, input_frames, prompt, height, width, num_inference_steps, strength, guidance_scale,
                                 negative_prompt, num_videos_per_prompt, eta, generator, latents, output_type, return_dict, callback,
                                 callback_steps, overlaps, multicontrolnetresiduals_pipeline, multicontrolnetresiduals_overlap_pipeline,
                                 num_images_per_prompt, clip_skip, cross_attention_kwargs, prompt_embeds, negative_prompt_embeds, epoch,
                                 output_dir, save_outputs, last_output_frames, use_lcm, lcm_origin_steps, guess_mode, ipa_scale,
                                 use_img2img, input_latents, last_output_latents, control_images, **kwargs)
        finally:
            if torch.get_num_threads() != host_threads:
                torch.set_num_threads(host_threads)

    def _denoise(self, video_length, input_frames, prompt, height, width, num_inference_steps, strength, guidance_scale,
                 negative_prompt, num_videos_per_prompt, eta, generator, latents, output_type, return_dict, callback,
                 callback_steps, overlaps, multicontrolnetresiduals_pipeline, multicontrolnetresiduals_overlap_pipeline,
                 num_images_per_prompt, clip_skip, cross_attention_kwargs, prompt_embeds, negative_prompt_embeds, epoch,
                 output_dir, save_outputs, last_output_frames, use_lcm, lcm_origin_steps, guess_mode, ipa_scale,
                 use_img2img, input_latents, last_output_latents, control_images, **kwargs):
        unet = self.unet
        sample_size = unet.config.get("sample_size") or 64
        height = height or sample_size * self.vae_scale_factor
        width = width or sample_size * self.vae_scale_factor
        if height % 8 or width % 8:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        device = self.device
        do_cfg = guidance_scale > 1.0
        if prompt_embeds is None or (do_cfg and negative_prompt_embeds is None):
            raise RuntimeError("pass prompt_embeds / negative_prompt_embeds (the CLIP text encoder is a 'next' component)")
        prompt_embeds = prompt_embeds.to(device).float()
        negative_prompt_embeds = None if negative_prompt_embeds is None else negative_prompt_embeds.to(device).float()
        if negative_prompt_embeds is None:
            negative_prompt_embeds = torch.zeros_like(prompt_embeds)

        # IP-Adapter tokens (:698-710)
        if self.ip_adapter is not None:
            fixed_tok, fixed_untok = kwargs.get("image_prompt_embeds"), kwargs.get("uncond_image_prompt_embeds")
            if fixed_tok is not None:
                # a FIXED image prompt (the tokens `get_image_embeds_4controlanimate` returns, computed once by the caller): the
                # window does not look at the previous window's output, so windows stay independent problems and can be
                # sharded (vid2vid.run_video_sharded).  The reference's animate() has these two parameters
                # (modules/controlanimate_pipeline.py:124-127) and never forwards them; here they are live.
                if fixed_untok is None:
                    raise ValueError("image_prompt_embeds needs uncond_image_prompt_embeds (both outputs of get_image_embeds_4controlanimate)")
                self.ip_adapter.set_scale(ipa_scale)
                prompt_embeds = torch.cat([prompt_embeds, fixed_tok.to(device).float().view(1, -1, prompt_embeds.shape[-1])], dim=1)
                negative_prompt_embeds = torch.cat([negative_prompt_embeds, fixed_untok.to(device).float().view(1, -1, prompt_embeds.shape[-1])], dim=1)
            elif last_output_frames is not None or kwargs.get("clip_image_embeds") is not None:
                img_tok, uncond_tok = self.ip_adapter.get_image_embeds_4controlanimate(
                    pil_image=None if last_output_frames is None else last_output_frames[0], scale=ipa_scale,
                    clip_image_embeds=kwargs.get("clip_image_embeds"))
                prompt_embeds = torch.cat([prompt_embeds, img_tok.to(device).float()], dim=1)
                negative_prompt_embeds = torch.cat([negative_prompt_embeds, uncond_tok.to(device).float()], dim=1)
            else:
                z = torch.zeros((1, 4, prompt_embeds.shape[-1]), device=device)
                prompt_embeds = torch.cat([prompt_embeds, z], dim=1)
                negative_prompt_embeds = torch.cat([negative_prompt_embeds, z], dim=1)
        lcm_prompt_embeds = prompt_embeds.contiguous()
        cfg_prompt_embeds = torch.cat([negative_prompt_embeds, prompt_embeds]).contiguous() if do_cfg else lcm_prompt_embeds

        # timesteps (:731-740)
        sched = self.scheduler
        native_lcm = use_lcm or (isinstance(sched, LCMScheduler) and not isinstance(sched, DiffusersLCMScheduler))
        if native_lcm:
            sched.set_timesteps(strength, num_inference_steps, lcm_origin_steps)
        else:
            sched.set_timesteps(num_inference_steps, device=device)
        first = 0
        if strength >= 1 or use_lcm:
            timesteps = sched.timesteps
        else:
            timesteps, num_inference_steps = self.get_timesteps(num_inference_steps, strength, device)
            first = len(sched.timesteps) - len(timesteps)

        if latents is None:
            latents = self.prepare_latents(input_frames, 1, unet.in_channels, video_length, height, width, torch.float32, device,
                                           generator, timesteps[:1], overlaps, strength, None, last_output_frames, use_lcm,
                                           use_img2img, input_latents, last_output_latents)
        latents = latents.to(device=device, dtype=torch.float32).contiguous()
        f = latents.shape[2]
        step_range = kwargs.get("step_range")
        whole = step_range is None or (step_range[0] == 0 and step_range[1] >= len(timesteps))  # the call covers the window's every step
        if step_range is not None:
            # parity-test hook ("teacher forcing"): run only loop iterations lo..hi-1 of the schedule, starting from the
            # given `latents` -- a step is then compared with the reference on IDENTICAL inputs
            lo, hi = step_range
            timesteps = timesteps[lo:hi]
            first += lo

        w_embedding = get_w_embedding(torch.tensor([float(guidance_scale)]), embedding_dim=256).to(device) if use_lcm else None

        cn = multicontrolnetresiduals_pipeline
        if cn is not None:
            ctrl = control_images if control_images is not None else input_frames
            cn.prep_control_images(ctrl, control_image_processor=None, epoch=epoch, output_dir=output_dir,
                                   save_outputs=save_outputs, do_classifier_free_guidance=do_cfg, guess_mode=guess_mode)
            if len(cn.prep_images[0]) % f:
                raise ValueError("number of control images must equal the number of frames")
        cn_single = use_lcm or guess_mode or not do_cfg  # (:811-813)
        cpad = unet.conv_in.cin_pad
        rep = 1 if use_lcm else (2 if do_cfg else 1)
        unet_prompt = lcm_prompt_embeds if use_lcm else cfg_prompt_embeds
        cn_prompt = lcm_prompt_embeds if cn_single else cfg_prompt_embeds
        denoised = None
        self.last_step_times = []
        self.eps_history = []
        use_graph = bool(self.use_hip_graph) and device.type == "cuda" and len(timesteps) > 1
        self.graph_replays = 0
        self.graph_fallback_reason = None
        self.graph_recapture_reason = None
        gs = None
        if not use_graph and self._graph_state is not None:
            # a call that does not replay (use_hip_graph off, a one-step call): let go of the captured graph and of what it pins
            # (the models, their weight arenas and caches, the control images) instead of holding GBs through the pipeline object
            self.release_graph()
        if use_graph:
            hh, ww = latents.shape[3], latents.shape[4]
            nets = list(getattr(cn, "controlnets", [])) if cn is not None else []
            for m_ in (unet, *nets):  # (the signature names the weight arenas: pack them now if this is the models' first use)
                m_._ensure_ready(device)
            sig = (id(unet), id(unet.arena), tuple(id(n.arena) for n in nets), id(cn), rep, f, hh, ww, cpad, str(unet.act_dtype),
                   tuple(unet_prompt.shape), tuple(cn_prompt.shape) if cn is not None else None, bool(cn_single), bool(guess_mode),
                   bool(use_lcm), w_embedding is not None, bool(getattr(self, "overlap_controlnet", True)), bool(getattr(self, "fuse_controlnet_adds", True)),
                   tuple(tuple(pi.shape) for pi in cn.prep_images) if cn is not None else None,
                   tuple(id(pi) for pi in cn.prep_images) if cn is not None else None,
                   # host-side scalars that are baked into captured launches
                   tuple(float(c) for c in cn.cond_scale) if cn is not None else None,
                   float(ipa_scale) if self.ip_adapter is not None else None)
            gs = self._graph_state
            if gs is not None and gs["sig"] != sig:
                self.graph_recapture_reason = "the call's signature changed at positions %s" % [i_ for i_, (a_, b_) in enumerate(zip(gs["sig"], sig)) if a_ != b_]
            if gs is None or gs["sig"] != sig:
                gs = self._graph_state = {
                    "sig": sig, "graph": None, "eps": None, "owned": None, "wgraphs": {}, "lat_in": None,
                    # the signature holds ids: keep the objects alive so that an id cannot be reused by another object
                    "keep": (unet, unet.arena, tuple(nets), tuple(n.arena for n in nets), cn,
                             tuple(cn.prep_images) if cn is not None else None),
                    "x": torch.empty((rep * f, hh, ww, cpad), device=device, dtype=unet.act_dtype),
                    "t": torch.zeros(1, device=device, dtype=torch.float32),
                    "unet_prompt": torch.empty_like(unet_prompt), "cn_prompt": torch.empty_like(cn_prompt),
                    "w": None if w_embedding is None else torch.empty_like(w_embedding)}
            # this window's conditioning goes INTO the tensors the (possibly already captured) kernels read
            gs["unet_prompt"].copy_(unet_prompt)
            gs["cn_prompt"].copy_(cn_prompt)
            if w_embedding is not None:
                gs["w"].copy_(w_embedding)
            unet_prompt, cn_prompt, w_embedding = gs["unet_prompt"], gs["cn_prompt"], gs["w"]
            captured = gs["graph"] is not None or bool(gs.get("wgraphs"))
            lost = self._graph_lost_model_caches(gs, unet, nets, cn) if captured else None
            if lost is not None:
                self.graph_recapture_reason = lost
            if lost is not None:
                # the caches the captured kernels read were replaced (a re-prepared model, more pipelines on one model than it keeps
                # cache slots for, a one-step call that let go of the graph): their buffers may be freed or stale
                logger.info("capturing the hipGraph again: %s", lost)
                gs["graph"] = gs["eps"] = gs["owned"] = None
                gs["wgraphs"] = {}
                captured = False
            if captured:  # a later window: the per-window caches (text / IP K/V, hint embeddings), in place
                unet.refresh_window_caches(gs["unet_prompt"])
                for k_, n_ in enumerate(nets):
                    n_.refresh_window_caches(gs["cn_prompt"], cn.prep_images[k_])

        def model_eps(x, tt):
            down = mid = None
            if cn is not None:
                x_cn = x if (rep == 1 or not cn_single) else x[:f]
                twice = rep == 2 and not cn_single  # the ControlNet sees both (identical) CFG halves of the latents
                if getattr(self, "overlap_controlnet", True):
                    # joined inside the UNet; the 13 residual adds fused into the zero convolutions when the batches agree
                    down = cn.residuals_nhwc_async(x_cn, tt, cn_prompt, guess_mode, cfg_identical_halves=twice,
                                                   fuse_images=x.shape[0] if getattr(self, "fuse_controlnet_adds", True) else 0)
                else:
                    down, mid = cn.residuals_nhwc(x_cn, tt, cn_prompt, guess_mode, cfg_identical_halves=twice)
            # x = latents_to_nhwc(latents, rep): for rep == 2 the two CFG halves are the same tensor (reference :797)
            return unet.forward_nhwc(x, rep, f, tt, unet_prompt, down, mid, timestep_cond=w_embedding, cfg_identical_halves=rep == 2)

        # The sampler's noise: the reference draws one CPU tensor per step inside the loop (:1601 `torch.randn`, global RNG;
        # diffusers' randn_tensor with the CPU generator) and uploads it.  The SAME values are drawn here in one go (one
        # `randn` of n x shape reproduces n consecutive draws when numel % 16 == 0 -- checked by tests/test_host_logic.py --
        # else n draws in a row) and uploaded once, while step 0's kernels run: no per-step host work on the critical path.
        needs_noise = bool(sched.needs_noise and len(sched.timesteps) > 1)
        native_noise = isinstance(sched, LCMScheduler) and not isinstance(sched, DiffusersLCMScheduler)
        noise_all = None

        def draw_all_noise():
            n_steps, shape = len(timesteps), tuple(latents.shape)
            gen = None if native_noise else generator
            st = self._noise_state
            need = (n_steps,) + shape
            if st is None or tuple(st["host"].shape) != need or st["dev"].device != device:
                host = torch.empty(need, dtype=torch.float32)
                try:
                    host = host.pin_memory()  # the upload is then asynchronous: queued behind step 0, the CPU moves on
                except RuntimeError:
                    pass
                st = self._noise_state = {"host": host, "dev": torch.empty(need, device=device, dtype=torch.float32), "done": None}
            if st["done"] is not None:
                st["done"].synchronize()  # (the previous window's upload has long finished)
            if latents.numel() % 16 == 0:
                torch.randn(need, generator=gen, dtype=torch.float32, out=st["host"])
            else:
                for k_ in range(n_steps):
                    torch.randn(shape, generator=gen, dtype=torch.float32, out=st["host"][k_])
            st["dev"].copy_(st["host"], non_blocking=True)
            if device.type == "cuda":
                st["done"] = torch.cuda.Event()
                st["done"].record()
            return st["dev"]

        pace_n = int(self.steps_in_flight) if device.type == "cuda" else 0
        pace = []  # blocking events, one per step in flight (oldest first)

        # ---- one replay per window ------------------------------------------------------------------------------------------
        window_mode = (use_graph and bool(self.window_graph) and callback is None and not self.record_eps and whole
                       and not getattr(sched, "multistep", False))
        if window_mode:
            plan = []
            for i, t in enumerate(timesteps):
                coef, clip = sched.coefficients(first + i)
                plan.append((float(t), float(sched.input_scale(first + i)), tuple(float(c) for c in coef), float(clip)))
            if needs_noise:
                noise_all = draw_all_noise()  # host draw + asynchronous upload into the static device buffer the graph reads
            g_eff = float(guidance_scale) if rep == 2 else 1.0
            wkey = (tuple(plan), g_eff, bool(use_lcm), None if noise_all is None else noise_all.data_ptr())
            if gs["lat_in"] is None or gs["lat_in"].shape != latents.shape:
                gs["lat_in"] = torch.empty_like(latents)
                gs["wgraphs"] = {}
            gs["lat_in"].copy_(latents)
            ent = gs["wgraphs"].get(wkey)
            if ent is None:
                try:
                    # warm-up outside the capture: one eager evaluation fills the models' per-window caches and the allocator pools
                    K.latents_to_nhwc(gs["lat_in"], cpad, rep, plan[0][1], unet.act_dtype, out=gs["x"])
                    gs["t"].fill_(plan[0][0])
                    model_eps(gs["x"], gs["t"])
                    torch.cuda.synchronize()
                    g_ = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g_, capture_error_mode="thread_local"):  # (another chain's thread may be launching meanwhile: chains.py)
                        lat_k, den_k = gs["lat_in"], None
                        for i, (t_k, scale_k, coef_k, clip_k) in enumerate(plan):
                            K.latents_to_nhwc(lat_k, cpad, rep, scale_k, unet.act_dtype, out=gs["x"])
                            gs["t"].fill_(t_k)
                            eps_k = model_eps(gs["x"], gs["t"])
                            lat_k, den_k = K.cfg_scheduler_step(eps_k, rep, g_eff, lat_k, noise_all[i] if needs_noise else None, coef_k, clip_k,
                                                                want_denoised=use_lcm)
                    ent = {"graph": g_, "out": lat_k, "den": den_k}
                    if len(gs["wgraphs"]) >= 2:  # (a pipeline alternating more than two schedules: keep the two newest)
                        gs["wgraphs"].pop(next(iter(gs["wgraphs"])))
                    gs["wgraphs"][wkey] = ent
                    gs["owned"] = self._model_cache_owners(gs, unet, nets, cn)
                    self.window_graph_fallback_reason = None
                except Exception as exc:  # never silent, never fatal: this window runs on the per-step path
                    ent = None
                    self.window_graph_fallback_reason = f"{type(exc).__name__}: {exc}"
                    logger.warning("capturing the whole-window hipGraph failed (%s); per-step graph instead", self.window_graph_fallback_reason)
                    torch.cuda.synchronize()
            if ent is not None:
                ent["graph"].replay()
                self.graph_replays += 1
                self.window_replays += 1
                ev = self._pace_event(0)
                ev.record()
                self._await(ev)  # (sleeps until the window is done: the caller's next device read would otherwise spin in the runtime)
                latents = ent["out"].clone()
                denoised = None if ent["den"] is None else ent["den"].clone()
                timesteps = timesteps[:0]  # the loop below has nothing left to do
        for i, t in enumerate(timesteps):
            idx = first + i
            in_scale = sched.input_scale(idx)
            if pace_n > 0 and len(pace) >= pace_n:
                self._await(pace.pop(0))
            if use_graph:
                K.latents_to_nhwc(latents, cpad, rep, in_scale, unet.act_dtype, out=gs["x"])   # [(rep f), h, w, 8]
                gs["t"].fill_(float(t))
                if gs["graph"] is None and i >= 1:  # caches and allocator pools are warm after the eager step 0
                    try:
                        torch.cuda.synchronize()
                        g_ = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g_, capture_error_mode="thread_local"):  # (another chain's thread may be launching meanwhile: chains.py)
                            gs["eps"] = model_eps(gs["x"], gs["t"])
                        gs["graph"] = g_
                        gs["owned"] = self._model_cache_owners(gs, unet, nets, cn)
                    except Exception as exc:  # capture is an optimisation only -- but never a silent one
                        use_graph = False
                        self._graph_state = None
                        self.graph_fallback_reason = f"{type(exc).__name__}: {exc}"
                        logger.warning("hipGraph capture failed (%s); this window runs eagerly", self.graph_fallback_reason)
                        torch.cuda.synchronize()
                if use_graph and gs["graph"] is not None:
                    gs["graph"].replay()
                    self.graph_replays += 1
                    eps = gs["eps"]
                else:
                    eps = model_eps(gs["x"], gs["t"])
            else:
                x = K.latents_to_nhwc(latents, cpad, rep, in_scale, unet.act_dtype)          # [(rep f), h, w, 8]
                eps = model_eps(x, t)
            if self.record_eps:
                self.eps_history.append(K.nhwc_to_ncfhw_f32(eps, rep, 4, f).cpu())
            noise = None
            if needs_noise:
                if noise_all is None:
                    noise_all = draw_all_noise()
                noise = noise_all[i]
            if getattr(sched, "multistep", False):
                # history-carrying samplers (DPM-Solver++, LMS, PNDM): CFG combine, then one linear-combination launch
                e = K.cfg_combined_eps(eps, rep, guidance_scale if rep == 2 else 1.0, latents)
                latents, den = sched.step_device(idx, e, latents, noise, K.lincomb), None
            else:
                coef, clip = sched.coefficients(idx)
                latents, den = K.cfg_scheduler_step(eps, rep, guidance_scale if rep == 2 else 1.0, latents, noise, coef, clip,
                                                    want_denoised=use_lcm)
            if use_lcm:
                denoised = den
            if pace_n > 0:
                ev = self._pace_event(i)
                ev.record()
                pace.append(ev)
            if callback is not None and i % callback_steps == 0:
                callback(i, t, latents)
        if pace:  # the caller's next device read (decode, .cpu()) would spin on the stream: sleep until the window is done
            self._await(pace[-1])
        final = denoised if use_lcm else latents
        if output_type == "latent" or self.vae is None:
            video = final
        else:
            video = self.decode_latents(final)
            if output_type == "tensor":
                video = torch.from_numpy(video)
        if not return_dict:
            return video
        return AnimationPipelineOutput(videos=video)
