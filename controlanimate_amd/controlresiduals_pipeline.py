"""MultiControlNetResidualsPipeline: the ControlNet stack of the denoising loop, on HIP kernels.

Same call surface as the reference's modules/controlresiduals_pipeline.py:
    MultiControlNetResidualsPipeline(hf_controlnet_names, cond_scale, use_lcm)            (:25-38)
    .prep_control_images(images, control_image_processor, epoch, output_dir, save_outputs,
                         do_classifier_free_guidance, guess_mode)  -> self.prep_images     (:226-273)
    .__call__(control_model_input[b,4,f,h,w], t, controlnet_prompt_embeds[b,L,768], frame_count,
              image_embeds=None, do_classifier_free_guidance=True, guess_mode=True)
        -> (tuple of 12 Tensor[b,C,f,h,w], Tensor[b,1280,f,h/8,w/8])                       (:278-316)
    attributes .controlnet (.nets, .half(), .dtype), .controlnets, .cond_scale, .controlnet_names.

The returned residuals are [b,C,f,h,w] VIEWS of channels-last storage, which the UNet consumes
without a copy (unet.py `_to_nhwc`).  Annotators (reference :97-150): canny is built in
(controlanimate_amd/annotators.py, a numpy restatement of cv2.Canny(img, 100, 200)); the learned detectors
(openpose / hed / lineart / mlsd / depth) are third-party models outside the loop and are NOT rebuilt: pass
already-annotated control images, or plug callables in through `annotators`.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import kernels as K
from .context import dispatch
from .controlnet import ControlNetModel, MultiControlNetModel


def _image_to_chw01(img) -> torch.Tensor:
    """PIL.Image / ndarray HWC uint8 / tensor -> float32 [3,H,W] in [0,1] (VaeImageProcessor with
    do_normalize=False, controlanimation_pipeline.py:160-163)."""
    if torch.is_tensor(img):
        t = img.float()
        if t.dim() == 4:
            t = t[0]
        return t
    arr = np.asarray(img)
    if arr.ndim == 2:
        arr = np.stack([arr] * 3, axis=-1)
    t = torch.from_numpy(arr[..., :3].copy()).permute(2, 0, 1).float()
    return t / 255.0 if arr.dtype == np.uint8 else t


class MultiControlNetResidualsPipeline:
    def __init__(self, hf_controlnet_names: Sequence[str], cond_scale: Sequence[float], use_lcm: bool,
                 controlnets: Optional[Sequence[ControlNetModel]] = None, device="cuda",
                 annotators: Optional[Dict[str, Callable]] = None):
        self.controlnet_names = list(hf_controlnet_names)
        if controlnets is None:
            # the reference's own constructor: ControlNetModel.from_pretrained(name) per name (:32-33) -- here from local
            # directories / the Hugging Face cache (there is no network; a missing model says where it was looked for)
            from .local_models import load_controlnet
            controlnets = [load_controlnet(n) for n in self.controlnet_names]
        if len(controlnets) != len(self.controlnet_names):
            raise ValueError("one ControlNetModel per name expected")
        self.controlnets = list(controlnets)
        self.controlnet = MultiControlNetModel(self.controlnets).to(device)
        self.cond_scale = list(cond_scale)
        self.use_lcm = use_lcm
        self.ip_adapter = None
        from .annotators import canny
        self.canny_processor = canny
        self.annotators = {"canny": canny, **dict(annotators or {})}
        self.prep_images: Optional[List[torch.Tensor]] = None
        self.device = torch.device(device)
        self.detect_identical_halves = True  # __call__ (the reference's own entry point) checks whether the two CFG halves of its input are equal
        self.lanes_used = 0  # HIP streams the ControlNet bodies of the last residuals_nhwc_async call were spread over

    # ------------------------------------------------------------------------------------------
    def prepare_controlnet_input_image(self, controlnet_model: str, image):
        if isinstance(image, torch.Tensor):
            return image  # tensors are control images that are already annotated / preprocessed
        for key, fn in self.annotators.items():
            if key in controlnet_model:
                return fn(image)
        return image  # already annotated

    def prep_control_images(self, images, control_image_processor=None, epoch=0, output_dir="tmp/output",
                            save_outputs=False, do_classifier_free_guidance=True, guess_mode=False):
        """images: list of f control images (PIL / arrays / tensors), or {controlnet name: list} when each
        net has its own pre-annotated set.  Result: self.prep_images[i] = Tensor[(b f),3,H,W] in [0,1];
        doubled for CFG exactly when the reference doubles it (:268-269)."""
        prep = []
        for name in self.controlnet_names:
            src = images[name] if isinstance(images, dict) else images
            frames = [_image_to_chw01(self.prepare_controlnet_input_image(name, im)) for im in src]
            ctrl = torch.stack(frames).to(self.device)
            doubled = bool(do_classifier_free_guidance and not guess_mode and not self.use_lcm)
            if doubled:
                ctrl = torch.cat([ctrl] * 2)
            old = self.prep_images[len(prep)] if self.prep_images is not None and len(self.prep_images) == len(self.controlnet_names) else None
            if old is not None and old.shape == ctrl.shape and old.dtype == ctrl.dtype and old.device == ctrl.device:
                # the next window's frames go INTO the tensor the ControlNets already know: their hint embeddings are then
                # refreshed in place (ControlNetModel.hint_embedding) and a captured hipGraph of the step stays valid
                old.copy_(ctrl)
                ctrl = old
            ctrl._cfg_doubled = doubled  # (both halves are the same frames: the hint embedding is computed for one; verified there)
            prep.append(ctrl)
        self.prep_images = prep

    # ------------------------------------------------------------------------------------------
    def residuals_nhwc(self, x_nhwc: torch.Tensor, t, controlnet_prompt_embeds: torch.Tensor, guess_mode: bool,
                       cfg_identical_halves: bool = False):
        """x_nhwc: [(b f),h,w,8] -> (12 NHWC residuals, mid), summed over the nets.  cfg_identical_halves: x_nhwc is one
        latent tensor repeated for the two CFG halves (see UNet3DConditionModel.forward_nhwc)."""
        if self.prep_images is None:
            raise RuntimeError("call prep_control_images() first")
        return self.controlnet.forward_nhwc(x_nhwc, t, controlnet_prompt_embeds, self.prep_images, self.cond_scale, guess_mode,
                                            cfg_identical_halves=cfg_identical_halves)

    def residuals_nhwc_async(self, x_nhwc: torch.Tensor, t, controlnet_prompt_embeds: torch.Tensor, guess_mode: bool,
                             cfg_identical_halves: bool = False, fuse_images: int = 0):
        """Same as residuals_nhwc, but enqueued on a second HIP stream so the ControlNet stack runs beside
        the UNet encoder (they are independent until the residual adds, unet.py:567-586): the kernels of
        one fill the CUs the other leaves idle in its launch tails and small-grid levels.  Returns
        join() -> (down, mid), which makes the CALLER's current stream wait for the residuals.
        UNet3DConditionModel.forward_nhwc accepts the join callable in place of `down_residuals`.

        fuse_images = the UNet's image count (rep * f).  When the ControlNets see the same batch, their zero convolutions are
        deferred: join(skips, mid) -- called by the UNet with its 12 skip tensors and its mid-block output once its encoder
        is done -- runs them on the second stream with those tensors as the epilogue's residual operand and returns
        (skips + residuals, mid + residual): the reference's 13 `sample + residual` adds (unet.py:567-576, 584-585) happen
        inside GEMMs that run anyway, instead of 13 extra passes over the tensors (`join.fuse`)."""
        if self.prep_images is None:
            raise RuntimeError("call prep_control_images() first")
        dev = x_nhwc.device
        main = torch.cuda.current_stream(dev)
        side = getattr(self, "_side_stream", None)
        if side is None or side.device != dev:
            side = self._side_stream = torch.cuda.Stream(device=dev)
        # (inside a hipGraph capture the fork/join below become graph edges; the allocator's cross-stream
        # bookkeeping is not needed there -- the graph's private pool outlives every replay)
        capturing = torch.cuda.is_current_stream_capturing()
        fuse = bool(fuse_images) and int(fuse_images) == int(x_nhwc.shape[0]) and self.prep_images is not None
        side.wait_stream(main)  # x_nhwc (and on the first call the weights) were produced on `main`
        # A stack of several ControlNets: their bodies are independent of each other (only the zero-convolution outputs are summed), so
        # they are spread over `dispatch.controlnet_streams` streams -- net i on stream i % n -- and meet again on `side`, which runs the
        # zero convolutions in the reference's order (MultiControlNetModel: residuals summed net by net).  Measured (one box, ms per step):
        # four nets (BASELINE config 3) 85.3 / 84.7 on one stream, 83.0 / 83.4 on two, 85.7 / 86.0 on four; two nets (config 4) 90.8 / 89.5
        # vs 90.6 / 90.5 -- so two streams, from three nets on.
        nets = len(getattr(self.controlnet, "nets", ()))
        n_par = max(1, min(int(dispatch.controlnet_streams), nets)) if nets >= 3 else 1
        self.lanes_used = n_par  # (what the last call really did: the tests of the multi-lane path assert on it)
        streams = [side]
        if n_par > 1:
            extra = getattr(self, "_extra_streams", None)
            if extra is None or len(extra) < n_par - 1 or extra[0].device != dev:
                extra = self._extra_streams = [torch.cuda.Stream(device=dev) for _ in range(n_par - 1)]
            streams += extra[: n_par - 1]
            for s_ in streams[1:]:
                s_.wait_stream(main)
        with torch.cuda.stream(side):
            bodies = self.controlnet.forward_bodies(x_nhwc, t, controlnet_prompt_embeds, self.prep_images, self.cond_scale, guess_mode,
                                                    cfg_identical_halves=cfg_identical_halves, streams=streams if n_par > 1 else None)
            for s_ in streams[1:]:
                side.wait_stream(s_)
            if not capturing and n_par > 1:
                for i, (outs_, xm_, _, _) in enumerate(bodies):
                    if i % n_par:  # produced on another stream, read by the zero convolutions on `side`
                        for t_ in (*outs_, xm_):
                            t_.record_stream(side)
            if not fuse:
                down, mid = self.controlnet.finish(bodies)
                done = side.record_event()
        if not capturing:
            for s_ in streams:
                x_nhwc.record_stream(s_)

        def join(skips=None, mid_x=None):
            cur = torch.cuda.current_stream(dev)
            if not fuse:
                cur.wait_event(done)
                if not capturing:
                    for r in (*down, mid):
                        r.record_stream(cur)  # allocated on `side`, read on `cur`
                return down, mid
            if skips is None or mid_x is None:
                raise RuntimeError("a fused ControlNet join needs the UNet's skip tensors and mid-block output")
            side.wait_stream(cur)  # the UNet's encoder (skips, mid) is complete on the caller's stream
            with torch.cuda.stream(side):
                d2, m2 = self.controlnet.finish(bodies, (list(skips), mid_x))
                fin = side.record_event()
            if not capturing:
                for s_ in (*skips, mid_x):
                    s_.record_stream(side)  # allocated on `cur`, read on `side`
            cur.wait_event(fin)
            if not capturing:
                for r in (*d2, m2):
                    r.record_stream(cur)
            return d2, m2

        join.fuse = fuse
        return join

    @torch.no_grad()
    def __call__(self, control_model_input: torch.Tensor, t, controlnet_prompt_embeds: torch.Tensor, frame_count: int,
                 image_embeds=None, do_classifier_free_guidance=True, guess_mode=True):
        b, c, f, h, w = control_model_input.shape
        if f != frame_count:
            raise ValueError("frame_count must equal the number of latent frames (SURVEY App. C-11)")
        net0 = self.controlnets[0]
        net0._ensure_ready(control_model_input.device)
        x = K.ncfhw_to_nhwc(control_model_input, net0.conv_in.cin_pad, net0.act_dtype)
        # The reference's loop hands over `torch.cat([latents] * 2)` under classifier-free guidance (controlanimation_pipeline.py:797-813):
        # when the two halves really are the same tensor twice (one small device compare per call), the ControlNets may treat them as
        # one problem (ControlNetModel.forward_body: shared prefix / the whole body once) -- what ControlAnimationPipeline tells them
        # directly.  A caller with different halves gets the all-images path, as before.
        same = bool(self.detect_identical_halves and do_classifier_free_guidance and not guess_mode and b == 2
                    and torch.equal(control_model_input[0], control_model_input[1]))
        down, mid = self.residuals_nhwc(x, t, controlnet_prompt_embeds, guess_mode, cfg_identical_halves=same)

        def to5(tn):  # [(b f),h,w,C] -> [b,C,f,h,w] view (no copy)
            bf, hh, ww, cc = tn.shape
            return tn.view(bf // frame_count, frame_count, hh, ww, cc).permute(0, 4, 1, 2, 3)

        return tuple(to5(d) for d in down), to5(mid)
