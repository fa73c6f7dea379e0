"""Host-side ControlNet annotators ("canny first").

`canny(image, low, high)` restates OpenCV's cv2.Canny(img, 100, 200) as the reference calls it
(modules/controlresiduals_pipeline.py:48-55: 8-bit RGB input, aperture 3, L1 gradient norm) in numpy:
3x3 Sobel with replicated borders; for colour input the channel with the largest |dx| + |dy| at each pixel;
non-maximum suppression with OpenCV's fixed-point tan(22.5 deg) sector test; hysteresis (strong > high,
weak > low connected through the 8-neighbourhood).  OpenCV is a THIRD-PARTY dependency that is absent here, so
this restatement is UNPINNED (no cv2 output to compare with); tests check the algorithmic properties.
The learned detectors (openpose, hed, lineart, mlsd, depth: controlnet_aux / transformers models, :56-61) are
not rebuilt: plug them in through `MultiControlNetResidualsPipeline(annotators={...})`.
"""
from __future__ import annotations

import numpy as np

try:
    from PIL import Image
except Exception:  # pragma: no cover
    Image = None


def _sobel(ch: np.ndarray):
    p = np.pad(ch.astype(np.int32), 1, mode="edge")
    dx = (p[:-2, 2:] + 2 * p[1:-1, 2:] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[1:-1, :-2] + p[2:, :-2])
    dy = (p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[:-2, 1:-1] + p[:-2, 2:])
    return dx, dy


def canny_edges(img: np.ndarray, low: float = 100, high: float = 200) -> np.ndarray:
    """img: [H,W] or [H,W,C] uint8 -> [H,W] uint8 edge map (0 / 255)."""
    a = np.asarray(img)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, c = a.shape
    dx = np.zeros((h, w), np.int32)
    dy = np.zeros((h, w), np.int32)
    mag = np.full((h, w), -1, np.int32)
    for k in range(c):  # the channel with the largest L1 gradient wins
        gx, gy = _sobel(a[:, :, k])
        m = np.abs(gx) + np.abs(gy)
        take = m > mag
        dx[take], dy[take], mag[take] = gx[take], gy[take], m[take]
    low_i, high_i = int(np.floor(low)), int(np.floor(high))
    # non-maximum suppression along the quantised gradient direction (OpenCV: TG22 = tan(22.5) * 2^15)
    pm = np.pad(mag, 1, mode="constant")
    ax, ay = np.abs(dx).astype(np.int64), np.abs(dy).astype(np.int64) << 15
    tg22x = ax * 13573
    tg67x = tg22x + (ax << 16)
    c0 = pm[1:-1, 1:-1]
    left, right = pm[1:-1, :-2], pm[1:-1, 2:]
    up, down = pm[:-2, 1:-1], pm[2:, 1:-1]
    horiz = ay < tg22x                      # gradient ~ horizontal: compare with left / right
    vert = ay > tg67x                       # ~ vertical: compare with up / down
    s = np.where((dx ^ dy) < 0, -1, 1)      # diagonal: sign decides which diagonal
    ul, dr = pm[:-2, :-2], pm[2:, 2:]
    ur, dl = pm[:-2, 2:], pm[2:, :-2]
    d1, d2 = np.where(s > 0, ul, ur), np.where(s > 0, dr, dl)
    keep = np.where(horiz, (c0 > left) & (c0 >= right), np.where(vert, (c0 > up) & (c0 >= down), (c0 > d1) & (c0 > d2)))
    cand = keep & (mag > low_i)
    strong = cand & (mag > high_i)
    # hysteresis: grow the strong set through weak candidates (8-neighbourhood) to a fixed point
    out = strong.copy()
    while True:
        p = np.pad(out, 1, mode="constant")
        nb = (p[:-2, :-2] | p[:-2, 1:-1] | p[:-2, 2:] | p[1:-1, :-2] | p[1:-1, 2:] | p[2:, :-2] | p[2:, 1:-1] | p[2:, 2:])
        grown = out | (cand & nb)
        if grown.sum() == out.sum():
            break
        out = grown
    return (out * 255).astype(np.uint8)


def canny(image, low: float = 100, high: float = 200):
    """The reference's canny_processor (:48-55): PIL RGB -> PIL RGB whose three channels are the edge map."""
    e = canny_edges(np.asarray(image), low, high)
    rgb = np.repeat(e[:, :, None], 3, axis=2)
    return Image.fromarray(rgb) if Image is not None and not isinstance(image, np.ndarray) else rgb
