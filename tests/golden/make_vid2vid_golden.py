"""Window-loop fixtures captured from the REFERENCE's own `vid2vid()` (scripts/vid2vid.py:31-292).  Container only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_vid2vid_golden.py      # writes tests/golden/vid2vid_loop.json

What runs as REFERENCE code: the whole function -- frame budget arithmetic (:62-79), the sliding-window loop (:168-268: overlap
re-feed, loop-back, strength override, the IP-Adapter "initial generation" double call, colour matching against the previous
window's reference frame, the cross-fade, the written-frame counter that decides how much of a window is emitted).
What is stood in (imports of scripts/vid2vid.py that cannot run here): omegaconf (a plain attribute bag), modules.upscaler
(never constructed: upscale = 1), modules.controlanimate_pipeline.ControlAnimatePipeline (a recording fake whose `animate`
returns frames that are a pure function of its inputs), modules.utils (FFMPEGProcessor = in-memory frame source / sink,
get_fps_frame_count_width_height = the scenario's numbers, match_colors = a recording transformation, video_to_high_fps = no-op),
time.sleep.  Frames are 64x64 RGB images of ONE grey value, so a frame is identified by that value.

The fixture holds, per scenario: the configuration, every animate() call (input values, last_output_frames values, strength,
overlaps, overlap, epoch, L, frame_count), every match_colors() call (reference value) and the values written to the encoder.
(No text-to-video scenario: with `input_video_path == ""` the reference itself stops at :168 with UnboundLocalError --
`intermediate_frame_count` is only assigned under `if has_input_video` -- so that mode has no reference behaviour to pin.)
tests/test_vid2vid_host.py replays the same scenarios through controlanimate_amd.vid2vid.run_windows with the same fakes."""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

SCENARIOS = {
    # name: (input frames or 0 for text-to-video, config overrides)
    "overlap2_loopback": (11, dict(frame_count=4, overlap_length=2, overlap_strength=0.6, loop_back_frames=1)),
    "no_overlap": (10, dict(frame_count=4, overlap_length=0)),
    "overlap3_no_loopback_cut": (30, dict(frame_count=8, overlap_length=3, overlap_strength=0.8, loop_back_frames=0, end_time="00:00:02")),
    "ip_initial_generation": (9, dict(frame_count=4, overlap_length=2, overlap_strength=0.5, do_initial_generation=1, use_ipadapter=1)),
    "ragged_tail": (7, dict(frame_count=4, overlap_length=1, overlap_strength=0.9)),
    "exact_multiple": (8, dict(frame_count=4, overlap_length=0)),
}
BASE = dict(input_video_path="in.mp4", output_video_dir="/tmp/ca_vid2vid_golden", save_frames=0, upscale=1, use_face_enhancer=0, upscale_first=0,
            start_time="00:00:00", end_time="00:01:00", fps=10, fps_ffmpeg=10, crf=23, ffmpeg_path="ffmpeg", width=64, height=64, seed=7,
            total_frames=0, strength=1.0, overlap_strength=1.0, loop_back_frames=0, do_initial_generation=0, use_ipadapter=0)


class Bag:
    """OmegaConf stand-in: attributes, None for a missing key (as a non-struct DictConfig)."""

    def __init__(self, d):
        self.__dict__.update(d)

    def __getattr__(self, k):
        return None


def gray(v):
    return Image.fromarray(np.full((64, 64, 3), int(v) % 256, np.uint8))


def val(img):
    return int(np.asarray(img)[0, 0, 0])


def animate_value(input_value, index, epoch, has_last):
    """The fake generator: a pure function of what animate() was given (shared with the replay test)."""
    base = input_value if input_value is not None else 3 * index
    return (base + 10 * epoch + (5 if has_last else 0)) % 256


def match_value(v, ref):
    return (v + (ref % 3)) % 256


def run_reference(n_frames, overrides):
    cfg = dict(BASE, **overrides)
    if n_frames == 0:
        cfg["input_video_path"] = ""
    rec = dict(config=dict(cfg), n_input_frames=n_frames, animate=[], match_colors=[], written=[])
    state = dict(read=0)

    class FFMPEGProcessor:
        def __init__(self, cmd, std_in=False, std_out=False):
            self.std_in, self.std_out = std_in, std_out

        def read(self, count):
            assert count == 64 * 64 * 3
            if state["read"] >= n_frames:
                return np.zeros((0,), np.uint8)
            state["read"] += 1
            return np.full((count,), state["read"] - 1, np.uint8)  # frame k has grey value k

        def write(self, arr):
            rec["written"].append(int(np.asarray(arr)[0, 0, 0]))

        def close(self):
            pass

    class ControlAnimatePipeline:
        def __init__(self, config):
            pass

        def animate(self, input_frames, last_output_frames, config, image_prompt_embeds=None, uncond_image_prompt_embeds=None):
            n = len(input_frames) if len(input_frames) else int(config.frame_count)
            rec["animate"].append(dict(inputs=[val(f) for f in input_frames], last=None if last_output_frames is None else [val(f) for f in last_output_frames],
                                       strength=float(config.strength), overlaps=int(config.overlaps), overlap=bool(config.overlap), epoch=int(config.epoch),
                                       L=int(config.L), frame_count=int(config.frame_count)))
            return [gray(animate_value(val(input_frames[i]) if len(input_frames) else None, i, int(config.epoch), last_output_frames is not None)) for i in range(n)]

    def match_colors(frames, ref):
        rec["match_colors"].append(val(ref))
        return [gray(match_value(val(f), val(ref))) for f in frames]

    mods = {
        "omegaconf": types.SimpleNamespace(OmegaConf=types.SimpleNamespace(load=lambda path: Bag(cfg), to_container=lambda c, resolve=True: dict(c.__dict__))),
        "modules": types.ModuleType("modules"),
        "modules.upscaler": types.SimpleNamespace(Upscaler=None),
        "modules.controlanimate_pipeline": types.SimpleNamespace(ControlAnimatePipeline=ControlAnimatePipeline),
        "modules.utils": types.SimpleNamespace(video_to_high_fps=lambda *a, **k: "done", FFMPEGProcessor=FFMPEGProcessor, match_colors=match_colors,
                                               get_fps_frame_count_width_height=lambda path: (10.0, n_frames, 64, 64)),
    }
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    import time as _time
    real_sleep = _time.sleep
    _time.sleep = lambda s: None
    try:
        src_path = os.path.join(REF, "scripts", "vid2vid.py")
        ns = {"__name__": "reference_vid2vid", "__file__": src_path}
        with open(src_path) as fh:
            exec(compile(fh.read(), src_path, "exec"), ns)  # (runs the reference file where it lies; nothing of it is stored)
        ns["vid2vid"]("unused.yaml")
    finally:
        _time.sleep = real_sleep
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return rec


if __name__ == "__main__":
    import contextlib
    import io
    out = {}
    for name, (n, ov) in SCENARIOS.items():
        with contextlib.redirect_stdout(io.StringIO()):
            out[name] = run_reference(n, ov)
        r = out[name]
        print(f"{name}: {len(r['animate'])} animate calls, {len(r['written'])} frames written, colour-match refs {r['match_colors']}")
    with open(os.path.join(HERE, "vid2vid_loop.json"), "w") as fh:
        json.dump(out, fh, indent=1)
