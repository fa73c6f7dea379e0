"""Host window orchestrator: the sliding-window loop of scripts/vid2vid.py:165-262
around `ControlAnimatePipeline.animate`, without ffmpeg / OmegaConf / the upscaler (frames come from and
go to any iterable / callable; the reference pipes raw RGB through ffmpeg, :93-136, :259-260).

Per window (file:line of the reference behaviour mirrored):
  :168-175  batch = the previous window's last `overlap_length` INPUT frames + (frame_count - overlaps) new ones
  :189-192  with overlaps: strength := overlap_strength; loop_back_frames feeds the previous OUTPUT frames back in
  :197-213  optional initial IP-Adapter round (run once, then re-run on its own last frames)
  :215-221  colour match every frame to `last_output_frame` (the reference calls the third-party color_matcher package
            with method 'hm-mkl-hm', modules/utils.py:116-130; `match_colors` below restates that compound:
            histogram matching -> Monge-Kantorovich linear transfer -> histogram matching)
  :93-136   raw-RGB frames through ffmpeg pipes (FFMPEGProcessor, modules/utils.py:87-113) -- `FFMPEGProcessor`,
            `ffmpeg_reader_cmd`, `ffmpeg_writer_cmd` below; `prefetch` decodes the next windows' frames on a host thread
            while the GPU denoises the current one (the reference reads synchronously)
  :221      last_output_frame = frames[overlap_length - 1]   (index -1, the last frame, when overlap_length == 0)
  :223-224  last_output_frames = frames[-overlap_length:]     (IP-Adapter image prompt / latent init of the next window)
  :226-228  cross-fade: frames[i] = blend(frames[i], previous_overlap_output[i], (n - i - 0.5) / n)
  :230-232  carry the blended tail and the matching input frames to the next window
  :235      emit len(batch) - overlaps frames, or the whole batch once the requested frame count is reached

Multi-GPU: with overlap_strength >= 1 and no IP-Adapter a window's denoising does not depend on
its predecessor's pixels, so windows can be denoised on different ranks (window_shard.windows_for_rank)
and only this host-side blend / colour-match pass runs in window order on rank 0: `blend_windows`.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Iterable, Iterator, List, Optional, Sequence

import numpy as np

try:  # PIL is what the reference's frames are; numpy HxWx3 uint8 arrays are accepted everywhere too
    from PIL import Image
except Exception:  # pragma: no cover
    Image = None


def _to_np(frame) -> np.ndarray:
    return np.asarray(frame)


def _like(arr: np.ndarray, proto):
    if Image is not None and not isinstance(proto, np.ndarray):
        return Image.fromarray(arr)
    return arr


def blend(a, b, alpha: float):
    """PIL.Image.blend semantics: a * (1 - alpha) + b * alpha, uint8 result of the type of `a`."""
    if Image is not None and not isinstance(a, np.ndarray) and not isinstance(b, np.ndarray):
        return Image.blend(a, b, alpha)
    x, y = _to_np(a).astype(np.float32), _to_np(b).astype(np.float32)
    out = x * (1.0 - alpha) + y * alpha
    return _like(np.clip(out, 0, 255).astype(np.uint8), a)  # PIL truncates


def match_colors_meanstd(frames: Sequence, ref_frame) -> List:
    """Cheap alternative hook: per-channel mean/std transfer to `ref_frame` (Reinhard).  The default is `match_colors`."""
    ref = _to_np(ref_frame).astype(np.float32)
    rm, rs = ref.reshape(-1, ref.shape[-1]).mean(0), ref.reshape(-1, ref.shape[-1]).std(0) + 1e-6
    out = []
    for fr in frames:
        x = _to_np(fr).astype(np.float32)
        m, s = x.reshape(-1, x.shape[-1]).mean(0), x.reshape(-1, x.shape[-1]).std(0) + 1e-6
        y = (x - m) / s * rs + rm
        out.append(_like(np.clip(y + 0.5, 0, 255).astype(np.uint8), fr))
    return out


def _hist_match(src: np.ndarray, ref: np.ndarray) -> np.ndarray:
    """Per-channel histogram matching (color_matcher `hm`): every source value is sent to the reference value of equal
    cumulative frequency (quantile mapping with linear interpolation).  float arrays [H,W,C]."""
    out = np.empty_like(src, dtype=np.float64)
    for c in range(src.shape[-1]):
        s, r = src[..., c].ravel(), ref[..., c].ravel()
        s_val, s_idx, s_cnt = np.unique(s, return_inverse=True, return_counts=True)
        r_val, r_cnt = np.unique(r, return_counts=True)
        s_q = np.cumsum(s_cnt).astype(np.float64) / s.size
        r_q = np.cumsum(r_cnt).astype(np.float64) / r.size
        out[..., c] = np.interp(s_q, r_q, r_val)[s_idx].reshape(src.shape[:-1])
    return out


def _mkl(src: np.ndarray, ref: np.ndarray) -> np.ndarray:
    """Monge-Kantorovich linear colour transfer (Pitie & Kokaram 2007; color_matcher `mkl`): the linear map that carries
    the source colour covariance onto the reference's, T = Cs^-1/2 (Cs^1/2 Cr Cs^1/2)^1/2 Cs^-1/2, about the means."""
    x = src.reshape(-1, src.shape[-1]).astype(np.float64)
    y = ref.reshape(-1, ref.shape[-1]).astype(np.float64)
    mx, my = x.mean(0), y.mean(0)
    cs, cr = np.cov(x, rowvar=False), np.cov(y, rowvar=False)
    eps = np.finfo(np.float64).eps

    def sqrtm(a):
        w, v = np.linalg.eigh(a)
        return (v * np.sqrt(np.clip(w, eps, None))) @ v.T

    cs_h = sqrtm(cs)
    cs_hi = np.linalg.inv(cs_h)
    t = cs_hi @ sqrtm(cs_h @ cr @ cs_h) @ cs_hi
    return ((x - mx) @ t + my).reshape(src.shape)


def _minmax(a: np.ndarray) -> np.ndarray:
    """color_matcher's `Normalizer.norm_fun`: (x - min) / (max - min) over the WHOLE array (all channels together);
    a constant array is returned unchanged."""
    lo, hi = float(a.min()), float(a.max())
    return (a - lo) / (hi - lo) if hi != lo else a


def match_colors(frames: Sequence, ref_frame, normalize: bool = True) -> List:
    """modules/utils.py:116-130: every frame is colour-matched to `ref_frame` with the compound 'hm-mkl-hm'
    (histogram matching, then the Monge-Kantorovich linear transfer, then histogram matching again) and written back
    as uint8.  The reference wraps the reference frame and every source frame in `Normalizer(...).type_norm()` and the
    result in `Normalizer(...).uint8_norm()` (:122, :125, :127) -- min-max normalisations of the `color_matcher`
    package: the transfer therefore targets the CONTRAST-STRETCHED reference frame and its result is stretched to the
    full 0..255 range again (`normalize=True`, the default; `False` = plain /255 and clip, the round-2 behaviour).
    `type_norm` keeps an integer image in its own type: the stretched uint8 frames are ROUNDED before the transfer (round 4:
    mirrored here; the first restatement kept them in float64).  The package is absent here: this restatement follows its
    published source and is unpinned against it; the property tests are in tests/test_vid2vid_host.py."""
    def type_norm(a: np.ndarray) -> np.ndarray:
        # Normalizer.type_norm(): an INTEGER image (PIL frames are uint8) comes back in its own type -- min-max stretched to the
        # type's range and ROUNDED, i.e. quantised before the transfer sees it; a float image is stretched to 0..1 unrounded.
        # (The transfer is scale-equivariant and the result is min-max normalised again, so only the rounding matters.)
        if not normalize:
            return a.astype(np.float64) / 255.0
        if np.issubdtype(a.dtype, np.integer):
            info = np.iinfo(a.dtype)
            return np.round(_minmax(a.astype(np.float64)) * (float(info.max) - float(info.min)) + float(info.min))
        return _minmax(a.astype(np.float64))

    ref = type_norm(_to_np(ref_frame))
    out = []
    for fr in frames:
        x = type_norm(_to_np(fr))
        y = _hist_match(_mkl(_hist_match(x, ref), ref), ref)
        if normalize:
            y = _minmax(y)
        out.append(_like(np.clip(np.round(y * 255.0), 0, 255).astype(np.uint8), fr))
    return out


class FFMPEGProcessor:
    """modules/utils.py:87-113: a child process whose stdin / stdout carries raw frames.  `read(count)` returns a uint8
    array of up to `count` bytes, `write(array)` sends its bytes, `close()` ends the input.  `cmd` is an argv list
    (ffmpeg_reader_cmd / ffmpeg_writer_cmd; no shell) or, as in the reference, a caller-supplied shell string."""

    def __init__(self, cmd, std_in: bool = False, std_out: bool = False):
        from subprocess import PIPE, Popen
        self.process = Popen(cmd, stdin=PIPE if std_in else None, stdout=PIPE if std_out else None, shell=isinstance(cmd, str))

    def read(self, count: int) -> np.ndarray:
        return np.frombuffer(self.process.stdout.read(count), dtype=np.uint8)

    def write(self, in_array) -> None:
        self.process.stdin.write(np.ascontiguousarray(in_array).tobytes())

    def close(self) -> int:
        if self.process.stdin is not None:
            self.process.stdin.close()
        return self.process.wait()


def ffmpeg_reader_cmd(path: str, width: int, height: int, fps: float, start_time: str = "00:00:00", end_time: Optional[str] = None,
                      ffmpeg_path: str = "ffmpeg") -> List[str]:
    """The decoder command of scripts/vid2vid.py:93-106: scaled raw rgb24 frames on stdout.  An argv LIST (run without a
    shell): a file name with quotes, `$()` or `;` in it stays a file name -- the reference interpolates it into a shell
    string."""
    cmd = [str(ffmpeg_path), "-loglevel", "error", "-ss", str(start_time)]
    if end_time:
        cmd += ["-to", str(end_time)]
    return cmd + ["-i", str(path), "-vf", f"fps={fps},scale={width}:{height}", "-f", "rawvideo", "-pix_fmt", "rgb24", "-"]


def ffmpeg_writer_cmd(path: str, width: int, height: int, fps: float, crf: int = 17, ffmpeg_path: str = "ffmpeg") -> List[str]:
    """The encoder command of scripts/vid2vid.py:120-136: raw rgb24 frames on stdin -> h264 (argv list, no shell)."""
    return [str(ffmpeg_path), "-loglevel", "error", "-y", "-f", "rawvideo", "-pix_fmt", "rgb24", "-s", f"{width}x{height}", "-r", str(fps),
            "-i", "-", "-c:v", "libx264", "-pix_fmt", "yuv420p", "-crf", str(crf), str(path)]


def frames_from_pipe(reader: FFMPEGProcessor, width: int, height: int) -> Iterator:
    """Raw rgb24 stream -> PIL frames (vid2vid.py:143-147,170-174: `read(width*height*3)` per frame until EOF)."""
    n = width * height * 3
    while True:
        buf = reader.read(n)
        if buf.size < n:
            return
        yield _like(buf.reshape(height, width, 3).copy(), None if Image is None else Image)


def prefetch(frames: Iterable, depth: int = 32) -> Iterator:
    """Decodes ahead on a host thread: the next windows' input frames are read (ffmpeg pipe, disk) while the GPU works
    on the current window.  Order-preserving; exceptions of the producer are re-raised in the consumer."""
    import queue
    import threading
    q: "queue.Queue" = queue.Queue(maxsize=depth)
    end = object()

    def work():
        try:
            for fr in frames:
                q.put(fr)
            q.put(end)
        except BaseException as exc:  # noqa: BLE001
            q.put(exc)

    threading.Thread(target=work, daemon=True).start()
    while True:
        item = q.get()
        if item is end:
            return
        if isinstance(item, BaseException):
            raise item
        yield item


@dataclass
class WindowConfig:
    """The per-window settings vid2vid.py writes into `config` before calling animate (:176-196)."""
    frame_count: int = 16
    overlap_length: int = 8
    strength: float = 1.0
    overlap_strength: float = 0.85
    loop_back_frames: bool = True
    do_initial_generation: bool = False
    # filled in per window
    overlap: bool = False
    overlaps: int = 0
    epoch: int = 0
    L: int = 0
    extra: dict = field(default_factory=dict)


AnimateFn = Callable[[List, Optional[List], WindowConfig], List]


def run_windows(input_frames: Optional[Iterable], animate: AnimateFn, cfg: WindowConfig, total_frames: Optional[int] = None,
                match_colors: Optional[Callable[[Sequence, object], List]] = match_colors) -> Iterator[List]:
    """Yields the output frames of each window, in order (what the reference writes to the encoder).
    input_frames: iterable of frames (vid2vid) or None (text-to-video: `total_frames` must be given and the
    batches are lists of None of the window length).  `animate(batch, last_output_frames, cfg)` is
    ControlAnimatePipeline.animate's contract (modules/controlanimate_pipeline.py:124-170)."""
    src = iter(input_frames) if input_frames is not None else None
    if src is None and total_frames is None:
        raise ValueError("text-to-video needs total_frames")
    base_strength = cfg.strength
    frame_count_cfg = cfg.frame_count
    overlap_frames: List = []
    overlap_inputs: List = []
    last_output_frames = None
    last_output_frame = None
    initial_done = not cfg.do_initial_generation
    emitted, epoch = 0, 0
    exhausted = False
    pending: List = []  # one frame of look-ahead: a stream's last window must be known when it is emitted

    def pull():
        if pending:
            return pending.pop()
        return next(src)

    # the reference counts written frames from 1 (:139) and stops once the count reaches the requested total
    while not exhausted and (total_frames is None or emitted + 1 < total_frames):
        batch: List = list(overlap_inputs)
        want = frame_count_cfg - len(overlap_frames)
        if src is not None:
            try:
                for _ in range(want):
                    batch.append(pull())
                pending.append(pull())
            except StopIteration:
                exhausted = True
            if len(batch) == len(overlap_inputs):  # no new frame arrived
                break
        else:
            batch += [None] * want
        cfg.overlap, cfg.overlaps = len(overlap_frames) > 0, len(overlap_frames)
        cfg.L = cfg.frame_count = len(batch)
        cfg.strength = base_strength
        if overlap_frames:
            cfg.strength = cfg.overlap_strength
            if cfg.loop_back_frames:
                batch[:len(overlap_frames)] = overlap_frames
        cfg.epoch = epoch
        epoch += 1
        if not initial_done:
            frames = animate(batch, last_output_frames, cfg)
            last_output_frame = frames[0]
            cfg.strength = cfg.overlap_strength
            cfg.overlaps = len(frames[-cfg.overlap_length:])
            frames = animate(batch, list(frames[-cfg.overlap_length:]), cfg)
            initial_done = True
        else:
            frames = animate(batch, last_output_frames, cfg)
        frames = list(frames)
        if last_output_frame is not None and match_colors is not None:
            frames = list(match_colors(frames, last_output_frame))
        last_output_frame = frames[max(cfg.overlap_length - 1, -1)]
        if cfg.overlap_length > 0:
            last_output_frames = frames[-cfg.overlap_length:]
        n = len(overlap_frames)
        for i, prev in enumerate(overlap_frames):
            frames[i] = blend(frames[i], prev, (n - i - 0.5) / n)
        if cfg.overlap_length > 0:
            overlap_frames = frames[-cfg.overlap_length:]
            overlap_inputs = batch[-cfg.overlap_length:]
        last = exhausted or (total_frames is not None and emitted + 1 + len(batch) >= total_frames)  # (:235)
        out_n = len(batch) if last else len(batch) - len(overlap_frames)
        emitted += out_n
        yield frames[:out_n]
        if last:
            break
    cfg.frame_count = frame_count_cfg


def blend_windows(windows: Sequence[Sequence], overlap_length: int,
                  match_colors: Optional[Callable[[Sequence, object], List]] = None) -> List:
    """Rank-0 pass of the window-sharded mode: `windows[k]` are the frames window k produced on some GPU
    (each window re-generated its first `overlap_length` frames); colour-match and cross-fade in window
    order exactly as the sequential loop does and return the final frame list."""
    out: List = []
    overlap_frames: List = []
    last_output_frame = None
    for k, frames in enumerate(windows):
        frames = list(frames)
        if last_output_frame is not None and match_colors is not None:
            frames = list(match_colors(frames, last_output_frame))
        last_output_frame = frames[max(overlap_length - 1, -1)]
        n = len(overlap_frames)
        for i, prev in enumerate(overlap_frames):
            frames[i] = blend(frames[i], prev, (n - i - 0.5) / n)
        if overlap_length > 0:
            overlap_frames = frames[-overlap_length:]
        is_last = k == len(windows) - 1
        out += frames if is_last else frames[:len(frames) - len(overlap_frames)]
    return out


def _run_windows_on_chains(pipe, plan, frames, cfg_of, ip_kw, rank, world, chains, device):
    """This rank's windows (r, r + W, ...) on `chains` facades over one set of models, one host thread and HIP stream each; the first
    window of every chain runs alone (eager step + hipGraph capture).  -> [(window index, uint8 frames [n, H, W, 3])]."""
    import threading
    import torch
    from . import window_shard as WS
    mine = WS.windows_for_rank(len(plan), rank, world)
    facades = [pipe] + [pipe.twin() for _ in range(chains - 1)]
    streams = [torch.cuda.Stream(device=device) for _ in range(chains)]
    results, errors = {}, [None] * chains

    def run(c, idxs):
        try:
            torch.cuda.set_device(device)
            with torch.cuda.stream(streams[c]):
                for k in idxs:
                    s, e = plan[k]
                    out = facades[c].animate(frames[s:e], None, cfg_of(k), **ip_kw)
                    results[k] = torch.from_numpy(np.stack([_to_np(f) for f in out]))
            streams[c].synchronize()
        except BaseException as exc:  # noqa: BLE001 -- re-raised in the caller's thread
            errors[c] = exc

    from .chains import one_host_thread
    with one_host_thread([fc.pipeline for fc in facades]):
        for c in range(min(chains, len(mine))):
            run(c, [mine[c]])
            if errors[c] is not None:
                raise errors[c]
        threads = [threading.Thread(target=run, args=(c, mine[c + chains::chains]), name=f"chain{c}", daemon=True) for c in range(chains)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    for e in errors:
        if e is not None:
            raise e
    return [(k, results[k]) for k in mine]


def run_video_sharded(config, frames: Sequence, components: Optional[dict] = None,
                      match_colors_fn: Optional[Callable[[Sequence, object], List]] = match_colors, device=None,
                      image_prompt_embeds=None, uncond_image_prompt_embeds=None, ip_reference_image=None,
                      single_rank_group: bool = False, chains_per_gpu: int = 1):
    """The window loop of scripts/vid2vid.py:168-268 over the GPUs of one node: one process per GPU (torchrun or any launcher
    that sets RANK / LOCAL_RANK / WORLD_SIZE), every rank calls this with the same `config` and the same input `frames`.

      1. `init_distributed` (RCCL over xGMI; gloo when CA_DIST_BACKEND=gloo);
      2. rank 0 builds ControlAnimatePipeline(config) from the checkpoints; every other rank builds the same object from
         the config files alone (`skeleton=True`: no weight file is read), packs it -- its arenas then have the final layout
         -- and RECEIVES the arena contents: `broadcast_weights(pipe.weight_buffers())`, a handful of multi-GB transfers;
      3. the sliding-window plan (`window_plan`); rank r runs `animate` on windows r, r + W, ... -- NO collective in the loop;
      4. the decoded frames of every window are gathered on rank 0 (uint8, ~12 MB per 16-frame 512x512 window), which does the
         colour matching against the previous window and the cross-fade of the overlap IN WINDOW ORDER (`blend_windows`),
         exactly as the sequential loop (`run_windows`) does, and returns the final frame list; the other ranks return None.

    Windows are independent problems -- and the result equal to the sequential loop's, frame for frame (tests/
    test_facade_gpu.py::test_sharded_video_equals_sequential) -- only when nothing flows from window k to window k + 1 before
    the blend: `overlap_strength >= 1` (no latent initialisation from the previous output, :566-604), `loop_back_frames` off
    (:197-199), no IP-Adapter conditioning on the previous window's frame (:698-710).  Any other configuration is a chain by
    construction and is refused here; run it with `run_windows` on one GPU.

    IP-Adapter (BASELINE config 4): the windows ARE independent when the image prompt is FIXED for the whole video instead of
    following the previous window.  Three sources, first match wins:
      * `image_prompt_embeds` / `uncond_image_prompt_embeds` -- the two parameters of the reference's animate()
        (modules/controlanimate_pipeline.py:124-127): the [1, 4, 768] token tensors of `get_image_embeds_4controlanimate`;
      * `ip_reference_image` -- a PIL image; rank 0 runs it through the CLIP vision encoder + ImageProjModel;
      * `do_initial_generation` in the config -- the reference's "initial round to acquire a baseline for the IP Adapter"
        (scripts/vid2vid.py:199-212): rank 0 generates window 0 once without an image prompt and takes its FIRST output frame
        as the baseline image (`last_output_frame = frames[0]`, :203).
    Whatever the source, the tokens are computed on rank 0 and broadcast (two 12 KB tensors), every window of every rank gets
    them through animate(image_prompt_embeds=...), and the result equals `run_windows` driven with the same fixed tokens
    (tests/test_facade_gpu.py::test_sharded_ip_adapter_video_equals_sequential).  With `use_ipadapter` and none of the three
    the windows form a chain and the call is refused as before.  (A sampler that draws its step noise
    from the SAME generator as the VAE's latent sampling -- diffusers' ancestral / LCM samplers -- sees a different generator
    state in the two modes, because the sequential loop also encodes the previous window's frames: statistically the same
    video, not the same bits.  Deterministic samplers and the native LCM sampler, whose noise comes from the global RNG that
    `animate` re-seeds per window, agree bit for bit.)

    chains_per_gpu > 1 (round 6): every rank keeps that many of ITS windows in flight on its GPU -- `ControlAnimatePipeline.twin()` facades
    over the same models, one host thread and HIP stream each (chains.py): the same frames, bit for bit, a few per cent more of them per
    second.  Not with the native LCM sampler (`use_lcm`), whose step noise comes from torch's global generator."""
    import torch
    from . import window_shard as WS
    from .controlanimate_pipeline import ControlAnimatePipeline, _get
    frame_count, overlap = int(_get(config, "frame_count", 16)), int(_get(config, "overlap_length", 0))
    if overlap > 0 and float(_get(config, "overlap_strength", 1.0)) < 1.0:
        raise ValueError("window sharding needs overlap_strength >= 1: with less, a window starts from the previous window's output "
                         "latents and the windows form a chain (use run_windows on one GPU)")
    if bool(_get(config, "loop_back_frames", False)) and overlap > 0:
        raise ValueError("window sharding needs loop_back_frames off: looped-back frames are the previous window's OUTPUT")
    use_ip = bool(_get(config, "use_ipadapter", 0))
    initial_round = use_ip and bool(_get(config, "do_initial_generation", False))
    if use_ip and image_prompt_embeds is None and ip_reference_image is None and not initial_round:
        raise ValueError("window sharding with the IP-Adapter needs a FIXED image prompt (image_prompt_embeds + uncond_image_prompt_embeds, "
                         "ip_reference_image, or do_initial_generation in the config): conditioned on the previous window's frame the "
                         "windows form a chain (use run_windows on one GPU)")
    rank, world, local_rank = WS.init_distributed(single_rank_group=single_rank_group)
    if device is None:
        device = torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1))
    if torch.device(device).type == "cuda":
        torch.cuda.set_device(device)
    pipe = ControlAnimatePipeline(config, components=components, device=device, skeleton=(rank != 0 and components is None))
    moved = WS.broadcast_weights(pipe.weight_buffers())
    frames = list(frames)
    plan = WS.window_plan(len(frames), frame_count, overlap)

    def cfg_of(k: int) -> dict:
        s, e = plan[k]
        c = dict(config) if isinstance(config, dict) else {kk: getattr(config, kk) for kk in dir(config) if not kk.startswith("_") and not callable(getattr(config, kk))}
        later = k > 0 and overlap > 0
        c.update(frame_count=e - s, L=e - s, epoch=k, overlap=later, overlaps=overlap if later else 0,
                 strength=float(_get(config, "overlap_strength", 1.0)) if later else float(_get(config, "strength", 1.0)))
        return c

    ip_kw = {}
    if use_ip:
        ip = pipe.pipeline.ip_adapter
        tok_shape = (1, ip.num_tokens, int(pipe.pipeline.unet.config.cross_attention_dim))
        both = torch.zeros((2,) + tok_shape, device=device, dtype=torch.float32)
        # the arguments are the same on every rank: a bad combination raises everywhere, before anybody waits in a collective
        if image_prompt_embeds is not None and uncond_image_prompt_embeds is None:
            raise ValueError("image_prompt_embeds needs uncond_image_prompt_embeds")
        # rank 0 computes the fixed prompt (an initial denoising round, a CLIP encode); if that fails, the other ranks must not sit in
        # the broadcast until the process-group timeout: a status word goes first and every rank raises together
        status = torch.zeros(1, device=device, dtype=torch.float32)
        failure = None
        if rank == 0:
            try:
                if image_prompt_embeds is not None:
                    tok, untok = image_prompt_embeds, uncond_image_prompt_embeds
                else:
                    image = ip_reference_image
                    if image is None:  # the reference's initial round (:199-203): window 0 without an image prompt, first frame = baseline
                        s0, e0 = plan[0]
                        image = pipe.animate(frames[s0:e0], None, cfg_of(0))[0]
                    tok, untok = ip.get_image_embeds_4controlanimate(pil_image=image, scale=float(_get(config, "ipa_scale", 0.4)))
                both[0].copy_(tok.to(device).float().view(tok_shape))
                both[1].copy_(untok.to(device).float().view(tok_shape))
            except Exception as exc:  # noqa: BLE001 -- re-raised below, on every rank
                failure = exc
                status.fill_(1.0)
        WS.broadcast_tensor(status, src=0)
        if float(status.item()) != 0.0:
            if failure is not None:
                raise failure
            raise RuntimeError("rank 0 failed while preparing the fixed IP-Adapter image prompt (its traceback has the cause)")
        WS.broadcast_tensor(both, src=0)
        ip_kw = dict(image_prompt_embeds=both[0], uncond_image_prompt_embeds=both[1])
        run_video_sharded.last_image_prompt = both.cpu()

    def run_window(k: int) -> torch.Tensor:
        s, e = plan[k]
        out = pipe.animate(frames[s:e], None, cfg_of(k), **ip_kw)
        return torch.from_numpy(np.stack([_to_np(f) for f in out]))  # [n, H, W, 3] uint8

    if int(chains_per_gpu) > 1:
        if bool(_get(config, "use_lcm", 0)):
            raise ValueError("chains_per_gpu > 1 needs a sampler that draws from its own generator: the native LCM sampler (use_lcm) reads "
                             "torch's global generator, which two windows in flight would interleave")
        windows = WS.gather_window_results(_run_windows_on_chains(pipe, plan, frames, cfg_of, ip_kw, rank, world, int(chains_per_gpu), device),
                                           len(plan))
    else:
        windows = WS.run_sharded(len(plan), run_window, rank, world)
    run_video_sharded.last_broadcast_bytes = moved
    if rank != 0 or windows is None:
        return None
    # (PIL frames, as the sequential loop handles them: Image.blend's arithmetic, byte for byte)
    as_frame = (lambda a: Image.fromarray(a)) if Image is not None else (lambda a: a)
    return blend_windows([[as_frame(w[i].numpy()) for i in range(w.shape[0])] for w in windows], overlap, match_colors_fn)
