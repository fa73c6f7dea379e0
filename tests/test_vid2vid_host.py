"""Host window orchestrator (controlanimate_amd/vid2vid.py) against a hand-worked run of the loop in
scripts/vid2vid.py:165-262 (frame_count 4, overlap_length 2, 8 input frames)."""
import numpy as np
from PIL import Image


def gray(v):
    return Image.fromarray(np.full((4, 4, 3), v, np.uint8))


def val(img):
    return int(np.asarray(img)[0, 0, 0])


def test_sequential_loop_matches_hand_worked_example():
    from controlanimate_amd.vid2vid import WindowConfig, run_windows
    calls = []

    def animate(batch, last_output_frames, cfg):
        calls.append(dict(inputs=[val(b) for b in batch], last=None if last_output_frames is None else [val(x) for x in last_output_frames],
                          strength=cfg.strength, overlaps=cfg.overlaps, epoch=cfg.epoch, L=cfg.L))
        return [gray(val(b) + 10 * cfg.epoch) for b in batch]

    cfg = WindowConfig(frame_count=4, overlap_length=2, strength=1.0, overlap_strength=0.6, loop_back_frames=True)
    out = [[val(f) for f in w] for w in run_windows([gray(i) for i in range(8)], animate, cfg, match_colors=None)]
    # window 0: inputs 0..3 -> outputs 0..3; emits 2, carries outputs [2,3] + inputs [2,3]
    # window 1: inputs = fed-back outputs [2,3] + new [4,5] -> [12,13,14,15]; cross-fade with [2,3] at 0.75 / 0.25
    #           -> [12*.25+2*.75, 13*.75+3*.25] = [4.5, 10.5] -> PIL truncates to [4, 10]; carries [14, 15]
    # window 2: inputs = [14,15] + [6,7] -> +20 -> [34,35,26,27]; fade with [14,15]: [34*.25+14*.75, 35*.75+15*.25] = [19, 30];
    #           the stream ends here, so the whole window is emitted
    assert out == [[0, 1], [4, 10], [19, 30, 26, 27]]
    assert [c["inputs"] for c in calls] == [[0, 1, 2, 3], [2, 3, 4, 5], [14, 15, 6, 7]]
    assert [c["last"] for c in calls] == [None, [2, 3], [14, 15]]
    assert [c["strength"] for c in calls] == [1.0, 0.6, 0.6] and [c["overlaps"] for c in calls] == [0, 2, 2]
    assert [c["epoch"] for c in calls] == [0, 1, 2] and all(c["L"] == 4 for c in calls)


def test_no_loop_back_refeeds_input_frames_and_total_frames_cutoff():
    from controlanimate_amd.vid2vid import WindowConfig, run_windows
    seen = []

    def animate(batch, last, cfg):
        seen.append([val(b) for b in batch])
        return [gray(val(b) + 100) for b in batch]

    cfg = WindowConfig(frame_count=4, overlap_length=2, loop_back_frames=False)
    out = list(run_windows((gray(i) for i in range(100)), animate, cfg, total_frames=6, match_colors=None))
    # the reference's written-frame counter starts at 1 (:139): 1 + 4 < 6 -> 2 frames; 3 + 4 >= 6 -> the whole window
    assert seen == [[0, 1, 2, 3], [2, 3, 4, 5]]
    assert [len(w) for w in out] == [2, 4]


def test_colour_match_reference_frame_and_text_to_video():
    from controlanimate_amd.vid2vid import WindowConfig, run_windows
    refs = []

    def mc(frames, ref):
        refs.append(val(ref))
        return list(frames)

    def animate(batch, last, cfg):
        return [gray(10 * cfg.epoch + i) for i in range(len(batch))]

    cfg = WindowConfig(frame_count=4, overlap_length=2)
    out = list(run_windows(None, animate, cfg, total_frames=6, match_colors=mc))
    # last_output_frame = frames[overlap_length - 1] AFTER colour matching, BEFORE the cross-fade (:221)
    assert refs == [1] and [len(w) for w in out] == [2, 4]
    cfg0 = WindowConfig(frame_count=3, overlap_length=0)
    refs.clear()
    out0 = list(run_windows(None, animate, cfg0, total_frames=6, match_colors=mc))
    assert refs == [2] and [len(w) for w in out0] == [3, 3]  # overlap 0: the reference frame is the last one (index -1)


def test_blend_windows_equals_sequential_for_independent_windows():
    from controlanimate_amd.vid2vid import WindowConfig, blend_windows, match_colors_meanstd, run_windows
    rng = np.random.default_rng(0)
    inputs = [Image.fromarray(rng.integers(0, 255, (8, 8, 3), dtype=np.uint8)) for _ in range(10)]

    def animate(batch, last, cfg):  # depends on its input frames only (overlap_strength >= 1, no IP-Adapter)
        return [Image.fromarray(255 - np.asarray(b)) for b in batch]

    cfg = WindowConfig(frame_count=4, overlap_length=2, loop_back_frames=False)
    seq = [f for w in run_windows(inputs, animate, cfg, match_colors=match_colors_meanstd) for f in w]
    from controlanimate_amd.window_shard import window_plan
    plan = window_plan(len(inputs), 4, 2)
    wins = [animate(inputs[a:b], None, cfg) for a, b in plan]          # any rank, any order
    par = blend_windows(wins, 2, match_colors=match_colors_meanstd)
    assert len(par) == len(seq) == 10
    assert all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(par, seq))


def test_meanstd_match_moves_statistics():
    from controlanimate_amd.vid2vid import match_colors_meanstd
    rng = np.random.default_rng(1)
    ref = rng.normal(100, 10, (16, 16, 3)).clip(0, 255).astype(np.uint8)
    src = rng.normal(150, 30, (16, 16, 3)).clip(0, 255).astype(np.uint8)
    out = np.asarray(match_colors_meanstd([src], ref)[0]).astype(np.float32)
    assert abs(out.mean() - ref.mean()) < 1.5 and abs(out.std() - ref.std()) < 1.5


def test_match_colors_hm_mkl_hm_properties():
    """The 'hm-mkl-hm' compound of modules/utils.py:116-130 (restated; the color_matcher package is absent)."""
    import numpy as np
    from controlanimate_amd.vid2vid import _hist_match, _mkl, match_colors
    rng = np.random.default_rng(0)
    ref = np.clip(rng.normal([150, 90, 60], [30, 20, 10], (48, 64, 3)), 0, 255).astype(np.uint8)
    src = np.clip(rng.normal([80, 120, 200], [15, 35, 25], (48, 64, 3)), 0, 255).astype(np.uint8)
    out = match_colors([src], ref, normalize=False)[0]
    assert out.dtype == np.uint8 and out.shape == src.shape
    # marginal statistics of every channel move onto the reference's (the final step is a histogram match)
    for c in range(3):
        assert abs(out[..., c].mean() - ref[..., c].mean()) < 1.5 and abs(out[..., c].std() - ref[..., c].std()) < 1.5
        assert np.abs(np.sort(out[..., c].ravel()).astype(float) - np.sort(ref[..., c].ravel())).mean() < 1.0
    # matching an image to itself changes nothing (up to one grey level of float round-off); MKL maps the covariance exactly
    assert np.abs(match_colors([ref], ref, normalize=False)[0].astype(int) - ref.astype(int)).max() <= 1
    # the default follows the reference's Normalizer wrappers (utils.py:122-127): the target is the min-max stretched
    # reference frame and the result spans 0..255
    dim = (ref.astype(np.float64) * 0.5 + 40).astype(np.uint8)  # a reference frame that does not span the range
    stretched = np.round((dim.astype(np.float64) - dim.min()) / (float(dim.max()) - float(dim.min())) * 255)
    outn = match_colors([src], dim)[0]
    assert outn.min() == 0 and outn.max() == 255
    for c in range(3):
        assert abs(outn[..., c].mean() - stretched[..., c].mean()) < 2.5
    # (<= 3: the MKL step perturbs equal grey levels by 1e-16, which splits their ties in the second histogram match)
    assert np.abs(match_colors([dim], dim)[0].astype(int) - stretched.astype(int)).max() <= 3
    # Normalizer.type_norm keeps an integer image in its type: stretched AND rounded before the transfer -- a uint8
    # frame and the same frame handed over as already-stretched-and-rounded values give the same result; a float frame with the
    # same values is stretched but not rounded (and may differ by a grey level)
    pre = np.round((dim.astype(np.float64) - dim.min()) / (float(dim.max()) - float(dim.min())) * 255).astype(np.uint8)
    assert pre.min() == 0 and pre.max() == 255
    assert np.array_equal(match_colors([src], dim)[0], match_colors([src], pre)[0])
    assert np.abs(match_colors([src], dim.astype(np.float32))[0].astype(int) - outn.astype(int)).max() <= 2
    x, y = src.astype(np.float64) / 255, ref.astype(np.float64) / 255
    t = _mkl(x, y).reshape(-1, 3)
    assert np.allclose(np.cov(t, rowvar=False), np.cov(y.reshape(-1, 3), rowvar=False), atol=1e-8)
    assert np.allclose(t.mean(0), y.reshape(-1, 3).mean(0), atol=1e-10)
    # histogram matching is monotone per channel: the rank order of the pixels is preserved
    h = _hist_match(x, y)
    i, j = rng.integers(0, x[..., 0].size, 200), rng.integers(0, x[..., 0].size, 200)
    a, b = x[..., 0].ravel(), h[..., 0].ravel()
    assert np.all((a[i] < a[j]) <= (b[i] <= b[j]))


def test_ffmpeg_pipe_class_and_prefetch():
    """FFMPEGProcessor (modules/utils.py:87-113) with `cat` standing for ffmpeg, frames_from_pipe and the async prefetcher."""
    import numpy as np
    from controlanimate_amd.vid2vid import FFMPEGProcessor, ffmpeg_reader_cmd, ffmpeg_writer_cmd, frames_from_pipe, prefetch
    w, h, n = 8, 6, 5
    rng = np.random.default_rng(1)
    frames = [rng.integers(0, 255, (h, w, 3), dtype=np.uint8) for _ in range(n)]
    proc = FFMPEGProcessor("cat", std_in=True, std_out=True)
    for f in frames:
        proc.write(f)
    proc.process.stdin.close()
    got = list(prefetch(frames_from_pipe(proc, w, h), depth=2))
    assert proc.process.wait() == 0
    assert len(got) == n and all(np.array_equal(np.asarray(a), b) for a, b in zip(got, frames))
    rd, wr = ffmpeg_reader_cmd('in "$(x)".mp4', 512, 512, 15, end_time="00:00:05"), ffmpeg_writer_cmd("out;rm.mp4", 512, 512, 15)
    assert isinstance(rd, list) and 'in "$(x)".mp4' in rd and "rawvideo" in rd and rd[rd.index("-to") + 1] == "00:00:05"
    assert isinstance(wr, list) and wr[-1] == "out;rm.mp4" and wr[wr.index("-s") + 1] == "512x512"
    echo = FFMPEGProcessor(["printf", "%s", "a;b"], std_out=True)  # an argv list runs without a shell
    assert bytes(echo.read(16)) == b"a;b" and echo.close() == 0

    def boom():
        yield 1
        raise ValueError("decoder died")
    import pytest
    with pytest.raises(ValueError, match="decoder died"):
        list(prefetch(boom()))


# ---- pinned to the REFERENCE's own loop: tests/golden/vid2vid_loop.json was captured from scripts/vid2vid.py::vid2vid() run
# behind stand-ins for ffmpeg / omegaconf / the pipeline (tests/golden/make_vid2vid_golden.py); the same fakes drive run_windows
import json
import os

import pytest

_GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vid2vid_loop.json")))


def _gray64(v):
    return Image.fromarray(np.full((64, 64, 3), int(v) % 256, np.uint8))


@pytest.mark.parametrize("name", sorted(_GOLD))
def test_run_windows_follows_the_reference_loop(name):
    """Every animate() call (inputs, last_output_frames, strength, overlaps, overlap flag, epoch, L, frame_count), every colour-match
    reference frame and every frame written to the encoder, in order, as scripts/vid2vid.py:168-268 produced them."""
    from controlanimate_amd.vid2vid import WindowConfig, run_windows
    g = _GOLD[name]
    c, n = g["config"], g["n_input_frames"]
    calls, refs = [], []

    def animate(batch, last, cfg):
        calls.append(dict(inputs=[val(f) for f in batch], last=None if last is None else [val(f) for f in last], strength=float(cfg.strength),
                          overlaps=int(cfg.overlaps), overlap=bool(cfg.overlap), epoch=int(cfg.epoch), L=int(cfg.L), frame_count=int(cfg.frame_count)))
        # (= make_vid2vid_golden.animate_value)
        return [_gray64((val(f) + 10 * cfg.epoch + (5 if last is not None else 0)) % 256) for f in batch]

    def match(frames, ref):
        refs.append(val(ref))
        return [_gray64((val(f) + val(ref) % 3) % 256) for f in frames]

    # the reference's frame budget (:62-79): fps x min(input duration, end - start), input duration = frames / input fps (10)
    def secs(t):
        h, m, s = (int(x) for x in t.split(":"))
        return 3600 * h + 60 * m + s
    budget = c["fps"] * min(n / 10.0, secs(c["end_time"]) - secs(c["start_time"]))
    wc = WindowConfig(frame_count=c["frame_count"], overlap_length=c["overlap_length"], strength=c["strength"], overlap_strength=c["overlap_strength"],
                      loop_back_frames=bool(c["loop_back_frames"]), do_initial_generation=bool(c["do_initial_generation"]))
    written = [val(f) for w in run_windows((_gray64(i) for i in range(n)), animate, wc, total_frames=budget, match_colors=match) for f in w]
    assert calls == g["animate"]
    assert refs == g["match_colors"]
    assert written == g["written"]
